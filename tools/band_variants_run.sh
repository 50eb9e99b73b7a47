for v in 0 1 2 4 16 20; do
  if [ $v = 0 ]; then lib=cfd_amd/liblbmhip.so; else lib=tools/ab/lbm_exp$v.so; fi
  echo "=== variant $v"
  LBMHIP_LIB=$PWD/$lib SWEEP_CAPS= SWEEP_LAGS= SWEEP_BAND_ROWS=8 SWEEP_BAND_CAPS=3,0 SWEEP_BAND_LAGS=0 timeout -k 10 120 python -u tools/fused_sweep.py 512 tgv FP32 2>&1 | grep -v "one launch" || exit 1
done
