#!/bin/bash
# Diagnostic: instruction-issue counters of k_collide_band (tools/band_stats.py, 6 steps per lag).
# GRBM_GUI_ACTIVE comes summed over the 8 XCDs, SQ_WAVE_CYCLES in units of 4 cycles.
# usage: [LBMHIP_LIB=...] [BAND_STATS_TOGETHER=2] tools/band_sq_pmc.sh <tag> <storage> <rows> "<lags>"
TAG=$1; export BAND_STATS_STORAGE=$2; ROWS=$3; LAGS=$4
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/bandsq_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/tools/band_stats.py 512 $ROWS 0 $LAGS > $OUT.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_collide_band" in k:
        acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for d, c in sorted(acc.items(), key=lambda kv: int(kv[0])):
    v = {k: sum(x) for k, x in c.items()}
    gui = v["GRBM_GUI_ACTIVE"]
    print("$TAG dispatch %s: cycles %.0f per XCD (%.3f ms at 2.4 GHz), waves %.0f, VALU instructions per wave %.0f, VALU issue busy %.1f %% of SIMD cycles (4 cycles per instruction, 1024 SIMDs), waves waiting %.1f %% of wave cycles, issuing %.1f %%, mean resident waves per SIMD %.2f"
          % (d, gui/8, gui/8/2.4e6, v["SQ_WAVES"], v["SQ_INSTS_VALU"]/v["SQ_WAVES"], 100.0*v["SQ_INSTS_VALU"]*4/(gui/8*1024),
             100.0*v["SQ_WAIT_INST_ANY"]/v["SQ_WAVE_CYCLES"], 100.0*v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_WAVE_CYCLES"]*4/(gui/8*1024)))
    print("   raw", {k: "%.4g" % x for k, x in v.items()})
PY
