"""Diagnostic: time of lbmhip_object_sum (kernel + read-back of the partial sums) on a 512^3 lattice with 2 M marked cells (GPU needed)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L
lib = cfd_amd.lib()
n = 512
for storage in ("FP32", "FP16S", "FP32"):
    dom = L.Domain(L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, L.STORAGE[storage], 0, L.FEATURES["FORCE_FIELD"], 0.01, 0))
    plane = np.zeros((n, n), dtype=np.uint8); plane[n // 4:n // 2, n // 4:n // 2] = 1
    for z in range(n // 4, n // 2):
        dom.upload(L.FIELD_FLAGS, plane, offset=z * n * n)
    L.check(lib.lbmhip_initialize(dom.h))
    s = (C.c_double * 4)()
    for what in (1, 0, 2):
        ts = []
        for r in range(12):
            t0 = time.perf_counter(); L.check(lib.lbmhip_object_sum(dom.h, what, C.c_uint8(1), None, None, s)); ts.append((time.perf_counter() - t0) * 1e3)
        print(storage, "what", what, " ".join("%.3f" % t for t in ts), "ms; cells", s[3], flush=True)
    dom.close()
