#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 kernel stats + PMC passes of bench.py) into profiles/<tag>_*.
usage: python tools/summarize_profile.py <tag> [workload-key]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)  # gpurun merges runs: older files of the same tag may remain
stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
bench_line = re.findall(r'\{"metric.*\}', open(os.path.join(src, "stats.log")).read())
bench = json.loads(bench_line[0]) if bench_line else {}

counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in [newest(os.path.join(d, "*", "*counter_collection.csv")) for d in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d)]:
    for r in csv.DictReader(open(f)):
        counters[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in counters.items():
    for c, v in cs.items():
        rows.append((k, c, len(v), sum(v) / len(v), min(v), max(v)))
with open(os.path.join(dst, "%s_pmc_summary.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
    for r in sorted(rows):
        w.writerow(r)

def mean(v):
    return sum(v) / len(v)


def traffic_of(names):
    """bytes per launch of the kernels `names` together.  gfx950: FETCH_SIZE counts 64 B per 128-B request, i.e. half the bytes
    (calibrated on known byte counts, profiles/r01_counter_calibration.csv); WRITE_SIZE is exact.  Both are reported in KiB."""
    fetch_kb = sum(mean(counters[k]["FETCH_SIZE"]) for k in names)
    write_kb = sum(mean(counters[k]["WRITE_SIZE"]) for k in names)
    d = {"kernels": names, "FETCH_SIZE_KiB": fetch_kb, "WRITE_SIZE_KiB": write_kb, "read_bytes": 2.0 * fetch_kb * 1024.0, "write_bytes": write_kb * 1024.0}
    d["bytes_per_launch"] = d["read_bytes"] + d["write_bytes"]
    if all("TCC_HIT_sum" in counters[k] for k in names):
        h, m = sum(mean(counters[k]["TCC_HIT_sum"]) for k in names), sum(mean(counters[k]["TCC_MISS_sum"]) for k in names)
        d["l2_hit_rate"] = h / (h + m)
    return d


out = {"tag": tag, "bench": bench}
commit = os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip() or "unknown"
tpath = os.path.join(dst, "traffic.json")
table = json.load(open(tpath)) if os.path.exists(tpath) else {}
# the step kernels of the run: a launch of two steps band by band is k_collide_band plus the k_collide_seam of the same instance
steps = []
for k in counters:
    m = re.match(r"void lbm::(k_collide_band|k_collide_pair|k_collide|k_collide_wide)<(.*)>\(", k)
    if not m or "FETCH_SIZE" not in counters[k] or "WRITE_SIZE" not in counters[k]:
        continue
    names = [k]
    if m.group(1) == "k_collide_band":
        names += [s for s in counters if s.startswith("void lbm::k_collide_seam<%s>" % m.group(2))]
    steps.append((len(counters[k]["FETCH_SIZE"]), m.group(1), m.group(2), names))
steps.sort(key=lambda e: -e[0])
out["step_kernels"] = []
storage_of = {"0": "fp32", "1": "fp16s", "2": "fp16c"}
for dispatches, kind, targs, names in steps:
    d = traffic_of(names)
    d["dispatches"] = dispatches
    out["step_kernels"].append(d)
    # bench.py's default run: Taylor-Green 512^3 D3Q19 SRT, the storage is the kernel's second template argument
    a = [v.strip() for v in targs.split(",")]
    if len(sys.argv) <= 2 and bench.get("config", {}).get("workload", "").startswith("taylor_green_512^3") and a[0] == "19" and kind in ("k_collide_band", "k_collide_pair") and dispatches >= 20:
        key = "taylor_green_512^3_per_gpu_d3q19_srt_%s_2steps%s" % (storage_of[a[1]], "_band" if kind == "k_collide_band" else "")
        table[key] = {"bytes_per_launch": round(d["bytes_per_launch"]), "kernel": " + ".join(names), "l2_hit_rate": d.get("l2_hit_rate"),
                      "source": "profiles/%s_pmc_summary.csv (rocprofv3 --pmc passes of bench.py; 2 x FETCH_SIZE + WRITE_SIZE, summed over the kernels of a launch; library at commit %s); "
                                "these counters sit between the L2 and the fabric, so bytes the Infinity Cache serves are counted like bytes from HBM, bytes the L2 serves are not" % (tag, commit)}
if steps and len(sys.argv) > 2:
    d = out["step_kernels"][0]
    table[sys.argv[2]] = {"bytes_per_launch": round(d["bytes_per_launch"]), "kernel": " + ".join(d["kernels"]),
                          "source": "profiles/%s_pmc_summary.csv (rocprofv3 --pmc passes; 2 x FETCH_SIZE + WRITE_SIZE; library at commit %s)" % (tag, commit)}
json.dump(table, open(tpath, "w"), indent=1, sort_keys=True)
json.dump(out, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
