"""Diagnostic: ms/step of stream_collide for the three storage formats at the library's default launch geometry (GPU needed).
usage: python tools/quick_bench.py [N] [Q]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import setups
from cfd_amd.lbm import check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 19
L = cfd_amd.lib()
for storage in ("FP32", "FP16S", "FP16C"):
    with cfd_amd.LBM((n, n, n), 0.01, storage=storage, Q=Q) as lbm:
        setups.taylor_green(lbm)
        lbm.run(100)  # covers the library's geometry search
        dom = lbm.domains[0]
        best = 1e9
        for _ in range(3):
            ms = C.c_float()
            check(L.lbmhip_timer_start(dom.h))
            for _ in range(40):
                lbm.do_time_step()
            check(L.lbmhip_timer_stop(dom.h, C.byref(ms)))
            best = min(best, ms.value / 40)
        B = lbm.bandwidth_bytes_per_cell()
        print("%-6s D3Q%d %d^3: %.3f ms/step  %.0f MLUPs/s  %.0f GB/s (%.1f%% of 8 TB/s)  [tuned=%d V=%d blocks/CU<=%d]" % (
            storage, Q, n, best, lbm.get_N() / best / 1e3, B * lbm.get_N() / best / 1e6, B * lbm.get_N() / best / 1e6 / 80,
            lbm.get_tuning("tuned"), lbm.get_tuning("cells_per_thread"), lbm.get_tuning("blocks_per_cu")), flush=True)
