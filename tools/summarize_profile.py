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

main = sorted((k for k in counters if "k_collide" in k), key=lambda k: -len(counters[k].get("FETCH_SIZE", [])))  # the step kernel has the most dispatches
out = {"tag": tag, "bench": bench}
if main:
    c = counters[main[0]]
    fetch_kb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    write_kb = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    # gfx950: FETCH_SIZE counts 64 B per 128-B request, i.e. half the bytes (calibrated below on known byte counts);
    # WRITE_SIZE is exact.  Both are reported in KiB.
    read_bytes, write_bytes = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
    out.update({"kernel": main[0], "FETCH_SIZE_KiB": fetch_kb, "WRITE_SIZE_KiB": write_kb, "hbm_read_bytes": read_bytes,
                "hbm_write_bytes": write_bytes, "hbm_bytes_per_launch": read_bytes + write_bytes})
    if "TCC_HIT_sum" in c:
        h, m = sum(c["TCC_HIT_sum"]) / len(c["TCC_HIT_sum"]), sum(c["TCC_MISS_sum"]) / len(c["TCC_MISS_sum"])
        out["l2_hit_rate"] = h / (h + m)
    key = sys.argv[2] if len(sys.argv) > 2 else bench.get("config", {}).get("workload")
    if key:
        tpath = os.path.join(dst, "traffic.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        t[key] = {"bytes_per_launch": round(read_bytes + write_bytes), "kernel": main[0],
                  "source": "profiles/%s_pmc_summary.csv (rocprofv3 --pmc passes of bench.py; 2 x FETCH_SIZE + WRITE_SIZE; library at commit %s)" % (
                      tag, os.popen("git -C %s rev-parse --short HEAD" % ROOT).read().strip() or "unknown")}
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
json.dump(out, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
