"""Diagnostic: step rate of lbmhip_stream_collide_steps (two time steps per launch) over its lag and residency cap,
against one launch per step.  (GPU needed)

usage: [SWEEP_CAPS=0,6,4,3] [SWEEP_LAGS=3,4,5,6,7,8] [SWEEP_BAND_ROWS=8,16] [SWEEP_BAND_LAGS=0,3,4] [SWEEP_BAND_CAPS=3,4,0] [SWEEP_BAND_BLOCK=128] [SWEEP_BAND_STEPS=2,3,4]
       python tools/fused_sweep.py [n] [config: tgv|channel] [storage] [cells collided together in the paired kernel: 1|2]
SWEEP_CAPS / SWEEP_LAGS empty: skip the sweep of k_collide_pair; SWEEP_BAND_ROWS (default empty) sweeps k_collide_band (lag 0 = automatic).
"""
import ctypes as C
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cfd_amd
from cfd_amd import lbm as L
from cfd_amd import setups

lib = cfd_amd.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
config = sys.argv[2] if len(sys.argv) > 2 else "tgv"
storage = sys.argv[3] if len(sys.argv) > 3 else "FP32"
together = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if config == "channel":
    kw = dict(N=(n, n, n), nu=0.01, features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0), collision="TRT", storage=storage)
else:
    kw = dict(N=(n, n, n), nu=0.01, storage=storage)
lbm = cfd_amd.LBM(**kw)
(setups.channel if config == "channel" else setups.taylor_green)(lbm)
lbm.initialize()
h = lbm.domains[0].h
B = lib.lbmhip_bandwidth_bytes_per_cell(19, L.STORAGE[storage], L.feature_mask(kw.get("features", ())))
f = [C.c_float(v) for v in kw.get("f", (0.0, 0.0, 0.0))]


def rate(steps):
    ms = C.c_float()
    L.check(lib.lbmhip_stream_collide_steps(h, C.c_uint64(lbm.t), steps, *f)); lbm.t += steps
    L.check(lib.lbmhip_timer_start(h))
    L.check(lib.lbmhip_stream_collide_steps(h, C.c_uint64(lbm.t), steps, *f)); lbm.t += steps
    L.check(lib.lbmhip_timer_stop(h, C.byref(ms)))
    per = ms.value / steps
    return per, n ** 3 / per / 1e3, B * n ** 3 / per / 1e6 / 80


lbm.set_tuning("fuse", 0)
for _ in range(20):
    rate(20)
    v = C.c_int()
    L.check(lib.lbmhip_get_tuning(h, b"tuned", C.byref(v)))
    if v.value:
        break
print("%s %s %d^3 one launch per step: %.3f ms/step %7.0f MLUPs/s %.1f %% of 8 TB/s" % ((config, storage, n) + rate(40)), flush=True)
lbm.set_tuning("fuse", 1)
lbm.set_tuning("pair_cells_together", together)
import os
def ints(name, default):
    return [int(v) for v in os.environ.get(name, default).split(",") if v.strip()]


caps = ints("SWEEP_CAPS", "0,6,4,3")
lags = ints("SWEEP_LAGS", "3,4,5,6,7,8")
for cap in caps:
    for lag in lags:
        lbm.set_tuning("pair_lag", lag); lbm.set_tuning("pair_blocks_per_cu", cap)
        print("%s %s %d^3 two steps per launch, lag %d, blocks/CU<=%d: %.3f ms/step %7.0f MLUPs/s %.1f %% of 8 TB/s" % ((config, storage, n, lag, cap) + rate(40)), flush=True)
lbm.set_tuning("fuse", 2)
lbm.set_tuning("pair_block", int(os.environ.get("SWEEP_BAND_BLOCK", "0")))  # lanes per workgroup (0 = automatic)
for rows in ints("SWEEP_BAND_ROWS", ""):
    for cap in ints("SWEEP_BAND_CAPS", "3,4,0"):
        for lag in ints("SWEEP_BAND_LAGS", "0,3,4,6"):
            for k in ints("SWEEP_BAND_STEPS", "0"):
                lbm.set_tuning("band_rows", rows); lbm.set_tuning("band_lag", lag); lbm.set_tuning("band_blocks_per_cu", cap); lbm.set_tuning("band_steps", k)
                r = rate(48)  # 48 = 24 launches of 2, 16 of 3, 12 of 4 steps
                print("%s %s %d^3 %d steps per launch band by band, %d rows, lag %d (%d), blocks/CU<=%d: %.3f ms/step %7.0f MLUPs/s %.1f %% of 8 TB/s, given up %d" %
                      ((config, storage, n, lbm.get_tuning("band_steps"), rows, lag, lbm.get_tuning("band_lag"), cap) + r + (lbm.get_tuning("band_given_up"),)), flush=True)
lbm.close()
