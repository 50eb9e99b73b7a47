# round-3 validation and measurement batch (GPU box); every step under its own time limit, stops at the first failure
set -x
export TMPDIR=/tmp
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3_gputests_final.log 2>&1 || { tail -30 gpurun_out/r3_gputests_final.log; exit 1; }
tail -3 gpurun_out/r3_gputests_final.log
timeout -k 10 300 python -u tools/fused_long_check.py FP32 1000 > gpurun_out/r3_long_fp32.log 2>&1 || { cat gpurun_out/r3_long_fp32.log; exit 1; }
timeout -k 10 300 python -u tools/fused_long_check.py FP16S 200 > gpurun_out/r3_long_fp16s.log 2>&1 || { cat gpurun_out/r3_long_fp16s.log; exit 1; }
cat gpurun_out/r3_long_fp32.log gpurun_out/r3_long_fp16s.log
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > gpurun_out/r3_bench2.log 2>&1 || { cat gpurun_out/r3_bench2.log; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench2_driver_flags.log 2>&1 || exit 1
timeout -k 10 300 python tools/slab_pair_bench.py > gpurun_out/r3_slab_pair2.log 2>&1 || { cat gpurun_out/r3_slab_pair2.log; exit 1; }
grep MLUPs gpurun_out/r3_slab_pair2.log
timeout -k 10 300 python tools/slab_pair_bench.py FP16S > gpurun_out/r3_slab_pair2_fp16s.log 2>&1 || exit 1
grep MLUPs gpurun_out/r3_slab_pair2_fp16s.log
