import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L
from oracle import harness as H
def run(n, storage="FP16S", V=0):
    small, steps, blob = 48, 6, 8
    cfg = L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, L.STORAGE[storage], 0, 0, 0.02, 0)
    dom = L.Domain(cfg); lib = cfd_amd.lib(); N = n**3
    L.check(lib.lbmhip_set_tuning(dom.h, b"cells_per_thread", V))
    base = n - small // 2 - blob // 2
    rng = np.random.RandomState(3)
    ux = (rng.rand(blob, blob, blob).astype(np.float32) - 0.5) * 0.1
    for dz in range(blob):
        for dy in range(blob):
            off = ((base + dz) * n + (base + dy)) * n + base
            dom.upload(L.FIELD_U, np.ascontiguousarray(ux[dz, dy]), offset=off * 4)
    L.check(lib.lbmhip_initialize(dom.h))
    for t in range(steps):
        L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(t), C.c_float(0), C.c_float(0), C.c_float(0)))
    L.check(lib.lbmhip_update_fields(dom.h, C.c_uint64(steps), C.c_float(0), C.c_float(0), C.c_float(0)))
    L.check(lib.lbmhip_sync(dom.h))
    o = n - small
    got = np.empty((3, small, small, small), dtype=np.float32); row = np.empty(small, dtype=np.float32)
    for c in range(3):
        for z in range(small):
            for y in range(small):
                dom.download(L.FIELD_U, row, offset=(c * N + ((o + z) * n + (o + y)) * n + o) * 4); got[c, z, y] = row
    dom.close()
    ref = H.OracleLBM((small,)*3, 0.02, storage=storage)
    b0 = small // 2 - blob // 2
    ref.u[0, b0:b0 + blob, b0:b0 + blob, b0:b0 + blob] = ux
    ref.run(steps); ref.read_from_device()
    d = got != ref.u
    print(n, storage, "V", V, "mismatch", int(d.sum()), "of", d.size, "maxdiff", float(np.abs(got - ref.u).max()), "max", float(np.abs(ref.u).max()), flush=True)
    if d.any():
        idx = np.argwhere(d); print(" first mismatches", idx[:5].tolist(), " bbox", idx.min(0).tolist(), idx.max(0).tolist())
run(256, "FP16S")
run(1600, "FP16S")
run(1600, "FP16S", 1)
