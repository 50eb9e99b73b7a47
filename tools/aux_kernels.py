"""Diagnostic: time of the kernels beside stream_collide on a 512^3 D3Q19 lattice (GPU needed)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L

lib = cfd_amd.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for storage in ("FP32", "FP16S"):
    fb = 4 if storage == "FP32" else 2
    cfg = L.Config(n, n, n + 2, 1, 1, 2, 0, 0, -1, 19, L.STORAGE[storage], 0, L.FEATURES["FORCE_FIELD"], 0.01, 0)
    dom = L.Domain(cfg)
    try:
        plane = np.zeros((n, n), dtype=np.uint8); plane[n // 4:n // 2, n // 4:n // 2] = 1
        for z in range(n // 4, n // 2):
            dom.upload(L.FIELD_FLAGS, plane, offset=z * n * n)
        N = dom.N

        def timed(label, call, bytes_moved, reps=5):
            call()
            ms = C.c_float()
            L.check(lib.lbmhip_timer_start(dom.h))
            for _ in range(reps):
                call()
            L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
            per = ms.value / reps
            print("%-6s %-22s %8.3f ms  %6.0f GB/s" % (storage, label, per, bytes_moved / per / 1e6), flush=True)

        zero = C.c_float(0)
        timed("initialize", lambda: L.check(lib.lbmhip_initialize(dom.h)), N * (19 * fb + 17))
        timed("stream_collide", lambda: L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(0), zero, zero, zero)), N * (38 * fb + 1 + 12))
        timed("update_fields", lambda: L.check(lib.lbmhip_update_fields(dom.h, C.c_uint64(0), zero, zero, zero)), N * (19 * fb + 17))
        timed("update_force_field", lambda: L.check(lib.lbmhip_update_force_field(dom.h, C.c_uint64(0))), N * 1 + (N // 16) * (19 * fb + 12))
        A = n * n
        timed("halo_pack fi (z)", lambda: L.check(lib.lbmhip_halo_pack(dom.h, L.HALO_FI, 2, C.c_uint64(0))), 2 * A * 5 * fb * 2)
        timed("halo_unpack fi (z)", lambda: L.check(lib.lbmhip_halo_unpack(dom.h, L.HALO_FI, 2, C.c_uint64(0))), 2 * A * 5 * fb * 2)
        s = (C.c_double * 4)()
        timed("object_sum force", lambda: L.check(lib.lbmhip_object_sum(dom.h, 1, C.c_uint8(1), None, None, s)), N * 1 + (N // 16) * 12)
    finally:
        dom.close()
