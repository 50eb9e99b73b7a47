"""Diagnostic: ms/step of stream_collide over cells-per-lane and residency caps for one storage format (GPU needed).
usage: python tools/tune.py [N] [storage] [Q] [caps, comma separated]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import setups
from cfd_amd.lbm import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
storage = sys.argv[2] if len(sys.argv) > 2 else "FP16S"
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 19
caps = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 2, 3, 4, 6]
L = cfd_amd.lib()
with cfd_amd.LBM((n, n, n), 0.01, storage=storage, Q=Q) as lbm:
    setups.taylor_green(lbm)
    lbm.run(4)
    dom = lbm.domains[0]
    B = lbm.bandwidth_bytes_per_cell()
    res = {}
    for rnd in range(3):
        for v in (1, 2, 4):
            for b in (128, 256):
                for k in caps:
                    lbm.set_tuning("cells_per_thread", v); lbm.set_tuning("block", b); lbm.set_tuning("blocks_per_cu", k)
                    lbm.run(2)
                    ms = C.c_float()
                    check(L.lbmhip_timer_start(dom.h))
                    for _ in range(20):
                        lbm.do_time_step()
                    check(L.lbmhip_timer_stop(dom.h, C.byref(ms)))
                    res.setdefault((v, b, k), []).append(ms.value / 20)
    for (v, b, k), ts in sorted(res.items(), key=lambda kv: sorted(kv[1])[1]):
        med = sorted(ts)[1]
        print("V=%d block=%3d blocks/CU<=%d  %.3f ms  %.0f MLUPs/s  %.0f GB/s (%.1f%%)" % (v, b, k, med, lbm.get_N() / med / 1e3, B * lbm.get_N() / med / 1e6, B * lbm.get_N() / med / 1e6 / 80), flush=True)
    lbm.read_from_device()
    print("finite", bool(np.isfinite(lbm.u).all()), "mean rho", float(lbm.rho.mean(dtype=np.float64)))
