"""Diagnostic: what the waits of k_collide_band cost (needs a build with LBM_BAND_EXP&32, tools/band_variants.sh 32).  (GPU needed)

usage: LBMHIP_LIB=$PWD/tools/ab/lbm_exp32.so LBMHIP_BAND_STATS=1 [BAND_STATS_STORAGE=FP16S] [BAND_STATS_TOGETHER=2] python tools/band_stats.py [n] [rows] [cap] [lags...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cfd_amd
from cfd_amd import setups

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lags = [int(v) for v in sys.argv[4:]] or [10, 32]
lbm = cfd_amd.LBM(N=(n, n, n), nu=0.01, storage=os.environ.get("BAND_STATS_STORAGE", "FP32"))
setups.taylor_green(lbm)
lbm.set_tuning("pair_cells_together", int(os.environ.get("BAND_STATS_TOGETHER", "1")))
lbm.set_tuning("fuse", 2); lbm.set_tuning("band_rows", rows); lbm.set_tuning("band_blocks_per_cu", cap)
for lag in lags:
    lbm.set_tuning("band_lag", lag)
    print("rows %d cap %d lag %d" % (rows, cap, lag), file=sys.stderr, flush=True)
    lbm.run(6)
lbm.close()
