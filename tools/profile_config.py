#!/usr/bin/env python3
"""Profiles one configuration of tools/config_bench.py with rocprofv3 on the GPU box and writes the tracked summary.

  python3 tools/profile_config.py <tag> <config-name> [more config_bench arguments]
  python3 tools/profile_config.py <tag> <update_fields|initialize>_<fp32|fp16s|fp16c> --bench=tools/aux_bench.py --kernel=<name part>

Passes (each its own process; counters never share a run with traces):
  1. plain run: the library's geometry search picks (cells per lane, cells together, residency cap, workgroup size)
  2. rocprofv3 --kernel-trace --stats with that geometry fixed  -> average duration of the step kernel
  3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* with that geometry fixed (three runs)
Output: gpurun_out/profiles/<tag>_<config>_kernel_stats.csv, _pmc_summary.csv, _summary.json (copied into profiles/ by hand);
raw files stay in gpurun_out/prof_<tag>_<config>/.

The summary's `frac` is algorithmic bytes per launch / average kernel duration / 8 TB/s; `traffic_ratio` is
(2 x FETCH_SIZE + WRITE_SIZE) / algorithmic bytes (gfx950 counts a 128-byte read request as 64 bytes);
`valu_issue_frac` is SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x clock): the share of the vector-ALU issue
slots the kernel filled (a wave64 vector instruction occupies its SIMD for four cycles).
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ = "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"
SIMDS, CLOCK_HZ = 1024, 2.4e9


def run(cmd, log, limit):
    with open(log, "w") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT, timeout=limit, env=dict(os.environ, TMPDIR="/tmp"))
    if r.returncode != 0:
        sys.stderr.write(open(log).read()[-3000:])
        raise SystemExit("%s failed with status %d" % (" ".join(cmd[:6]), r.returncode))
    return open(log).read()


def main():
    tag, name, extra = sys.argv[1], sys.argv[2], [a for a in sys.argv[3:] if not a.startswith(("--bench=", "--kernel="))]
    script = next((a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("--bench=")), "tools/config_bench.py")
    kernel_part = next((a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("--kernel=")), "k_collide")
    out = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, name))
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    bench = ["python3", script, name] + extra
    text = run(bench, os.path.join(out, "tune.log"), 600)
    m = re.search(r"(\d+) MLUPs/s +(\d+) B/cell.*\[tuned=(\d) V=(\d+) P=(\d+) blocks/CU<=(-?\d+) block=(\d+)\]", text)
    if not m:
        raise SystemExit("no result line in:\n" + text[-2000:])
    mlups_tuned, B = float(m.group(1)), int(m.group(2))
    geometry = "%s,%s,%s,%s" % (m.group(4), m.group(5), m.group(6), m.group(7))
    dims = re.search(r"(\d+)x *(\d+)x *(\d+)", text)
    cells = int(dims.group(1)) * int(dims.group(2)) * int(dims.group(3))
    fixed = bench + ["--geometry=" + geometry, "--steps=20"]
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(out, "stats"), "--"] + fixed, os.path.join(out, "stats.log"), 600)
    for label, counters in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq", SQ)):
        run(["rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", os.path.join(out, "pmc_" + label), "--"] + fixed,
            os.path.join(out, "pmc_%s.log" % label), 600)

    dst = os.path.join(ROOT, "gpurun_out", "profiles")  # only gpurun_out/ travels back from the GPU box; copy into profiles/ afterwards
    os.makedirs(dst, exist_ok=True)
    prefix = os.path.join(dst, "%s_%s" % (tag, name))
    newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
    stats_csv = newest(os.path.join(out, "stats", "*", "*kernel_stats.csv"))
    shutil.copy(stats_csv, prefix + "_kernel_stats.csv")
    kernels = [r for r in csv.DictReader(open(stats_csv)) if kernel_part in r["Name"]]
    step = max(kernels, key=lambda r: int(r["Calls"]))
    avg_ns = float(step["AverageNs"])

    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob(os.path.join(out, "pmc_*")):
        if os.path.isdir(d):
            for r in csv.DictReader(open(newest(os.path.join(d, "*", "*counter_collection.csv")))):
                counters[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(prefix + "_pmc_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
        for k in sorted(counters):
            for c in sorted(counters[k]):
                v = counters[k][c]
                w.writerow((k, c, len(v), sum(v) / len(v), min(v), max(v)))
    c = counters[step["Name"]]
    mean = lambda key: sum(c[key]) / len(c[key]) if c.get(key) else None
    algorithmic = float(B) * cells
    summary = {"tag": tag, "config": name, "kernel": step["Name"], "geometry_V_P_blocksPerCU_block": geometry, "calls": int(step["Calls"]),
               "avg_kernel_ns": avg_ns, "cells": cells, "bytes_per_cell": B, "algorithmic_bytes_per_launch": algorithmic,
               "achieved_GBs": algorithmic / avg_ns, "frac": algorithmic / avg_ns / 8000.0, "mlups_from_kernel_time": cells / avg_ns * 1e3,
               "mlups_untraced_run": mlups_tuned}
    if mean("FETCH_SIZE") is not None and mean("WRITE_SIZE") is not None:
        traffic = 2.0 * mean("FETCH_SIZE") * 1024.0 + mean("WRITE_SIZE") * 1024.0
        summary.update({"FETCH_SIZE_KiB": mean("FETCH_SIZE"), "WRITE_SIZE_KiB": mean("WRITE_SIZE"), "hbm_bytes_per_launch": traffic, "traffic_ratio": traffic / algorithmic})
    if mean("SQ_INSTS_VALU") is not None:
        valu, waves = mean("SQ_INSTS_VALU"), mean("SQ_WAVES")
        summary.update({"SQ_INSTS_VALU": valu, "SQ_INSTS_SALU": mean("SQ_INSTS_SALU"), "SQ_WAVES": waves,
                        "valu_insts_per_wave": valu / waves if waves else None, "valu_insts_per_cell": valu * 64.0 / cells,
                        "valu_issue_frac": valu * 4.0 / (SIMDS * avg_ns * 1e-9 * CLOCK_HZ),
                        "SQ_ACTIVE_INST_VALU": mean("SQ_ACTIVE_INST_VALU"), "SQ_ACTIVE_INST_VMEM": mean("SQ_ACTIVE_INST_VMEM"),
                        "SQ_WAVE_CYCLES": mean("SQ_WAVE_CYCLES"), "SQ_BUSY_CYCLES": mean("SQ_BUSY_CYCLES"), "SQ_BUSY_CU_CYCLES": mean("SQ_BUSY_CU_CYCLES")})
    json.dump(summary, open(prefix + "_summary.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
