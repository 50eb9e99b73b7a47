for st in FP16S FP16C; do for tog in 1 2; do
  echo "== $st cells together $tog"
  SWEEP_CAPS= SWEEP_LAGS= SWEEP_BAND_ROWS=0,32 SWEEP_BAND_CAPS=0,7,6 SWEEP_BAND_LAGS=0,3,5 timeout -k 10 280 python -u tools/fused_sweep.py 512 tgv $st $tog || exit 1
done; done
