#!/bin/bash
# Profiles the default bench.py workload with rocprofv3 on the GPU box: one kernel-trace/stats run and separate
# PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).  Results land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01}
ARGS=${2:-"--no-cpu-baseline"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $R/bench.py $ARGS > $OUT/pmc_l2.log 2>&1
find $OUT -name "*.csv" | head -20
