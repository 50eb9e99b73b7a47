"""Diagnostic: update_fields and initialize at 512^3 D3Q19 over cells per lane and residency caps (GPU needed)."""
import ctypes as C
import sys


sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L

lib = cfd_amd.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
zero = C.c_float(0)
for storage in ("FP32", "FP16S", "FP16C"):
    fb = 4 if storage == "FP32" else 2
    dom = L.Domain(L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, L.STORAGE[storage], 0, 0, 0.01, 0))
    try:
        N = dom.N
        L.check(lib.lbmhip_set_tuning(dom.h, b"autotune", 0))

        def timed(label, call, bytes_moved, reps=10):
            call(); call()
            ms = C.c_float()
            L.check(lib.lbmhip_timer_start(dom.h))
            for _ in range(reps):
                call()
            L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
            per = ms.value / reps
            print("%-6s %-34s %8.3f ms  %6.0f GB/s  %5.1f %%" % (storage, label, per, bytes_moved / per / 1e6, bytes_moved / per / 1e6 / 80), flush=True)

        B = 19 * fb + 17
        timed("initialize", lambda: L.check(lib.lbmhip_initialize(dom.h)), N * B)
        for V in ((1, 2) if storage == "FP32" else (1, 2, 4)):
            for cap in (0, 2, 3, 4, 6, 8):
                L.check(lib.lbmhip_set_tuning(dom.h, b"cells_per_thread", V))
                L.check(lib.lbmhip_set_tuning(dom.h, b"blocks_per_cu", cap))
                timed("update_fields V=%d blocks/CU<=%d" % (V, cap), lambda: L.check(lib.lbmhip_update_fields(dom.h, C.c_uint64(0), zero, zero, zero)), N * B)
    finally:
        dom.close()
