set -x
for name in channel512_trt_sgs_fp16s channel512_trt_sgs gui512_fp16s q27_fp16c_force_slab tgv512_fp32_sgs; do
  timeout -k 10 280 python -u tools/config_bench.py $name || exit 1
  timeout -k 10 280 python -u tools/config_bench.py --fast-div $name || exit 1
done
