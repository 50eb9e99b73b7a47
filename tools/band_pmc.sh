#!/bin/bash
# Diagnostic: FETCH_SIZE / WRITE_SIZE of k_collide_band at several lags (tools/band_stats.py runs 6 steps per lag), optionally with a
# variant build.  usage: tools/band_pmc.sh <tag> <rows> <cap> "<lags>" [library]
TAG=$1; ROWS=$2; CAP=$3; LAGS=$4; LIBF=$5
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
[ -n "$LIBF" ] && export LBMHIP_LIB=$R/$LIBF
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
	OUT=$R/gpurun_out/bandpmc_$TAG/$c
	mkdir -p $OUT
	timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT -- python3 $R/tools/band_stats.py 512 $ROWS $CAP $LAGS > $OUT.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/bandpmc_$TAG/%s/*/*counter_collection.csv" % c)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_collide_band" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    # 3 launches per lag, in the order of the lags
    lags = "$LAGS".split()
    for i, lag in enumerate(lags):
        v = vals[3*i:3*i+3]
        gb = sum(v)/len(v)*1024*(2 if c == "FETCH_SIZE" else 1)/1e9
        print("$TAG rows $ROWS cap $CAP lag %s %s: %.2f GB per launch (k_collide_band only)" % (lag, c, gb))
PY
