#!/bin/bash
# usage: pmc_cfg.sh <config-name>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/config_bench.py $1 > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/config_bench.py $1 > $OUT/write.log 2>&1
tail -1 $OUT/write.log
