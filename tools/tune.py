"""Interleaved A/B timing of launch geometries of the stream_collide kernel in ONE process (GPU needed).
usage: python tools/tune.py [N] [storage] [rounds]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import setups
from cfd_amd.lbm import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
storage = sys.argv[2] if len(sys.argv) > 2 else "FP32"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 19
steps = 20
lbm = cfd_amd.LBM((n, n, n), 0.01, storage=storage, Q=Q)
t0 = time.time()
lbm.rho[...] = 1.0
x = np.arange(n, dtype=np.float32)
lbm.u[0] = 0.1 * np.sin(2 * np.pi * x / n)[None, :, None] * np.cos(2 * np.pi * x / n)[:, None, None] * np.ones(n, dtype=np.float32)[None, None, :]
lbm.run(4)
print("init %.1fs" % (time.time() - t0), flush=True)
L = cfd_amd.lib()
dom = lbm.domains[0]
B = lbm.bandwidth_bytes_per_cell()
results = {}
variants = [(v, b, k) for v in (1, 2, 4) for b in (128, 256) for k in (-1, 0, 3)]
for r in range(rounds):
    for v, b, k in variants:
        lbm.set_tuning("cells_per_thread", v)
        lbm.set_tuning("block", b)
        lbm.set_tuning("blocks_per_cu", k)
        lbm.run(2)
        ms = C.c_float()
        check(L.lbmhip_timer_start(dom.h))
        for _ in range(steps):
            lbm.do_time_step()
        check(L.lbmhip_timer_stop(dom.h, C.byref(ms)))
        results.setdefault((v, b, k), []).append(ms.value / steps)
for (v, b, k), ts in sorted(results.items()):
    best, med = min(ts), sorted(ts)[len(ts) // 2]
    print("V=%d block=%3d blocks/CU<=%d  ms/step min %.3f med %.3f  MLUPs %.0f  GB/s %.0f (%.1f%% of 8 TB/s)" % (
        v, b, k, best, med, lbm.get_N() / med / 1e3, B * lbm.get_N() / med / 1e6, 100 * B * lbm.get_N() / med / 1e6 / 8000), flush=True)
lbm.read_from_device()
print("finite", bool(np.isfinite(lbm.u).all()), "mean rho", float(lbm.rho.mean(dtype=np.float64)))
