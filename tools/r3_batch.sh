set -x
export TMPDIR=/tmp
R=$PWD
cd /tmp
SLAB_ONLY_PAIRED=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_slab -- python3 $R/tools/slab_pair_bench.py > $R/gpurun_out/prof_r03_slab.log 2>&1 || { tail -20 $R/gpurun_out/prof_r03_slab.log; exit 1; }
find $R/gpurun_out/prof_r03_slab -name "*.csv" | head
