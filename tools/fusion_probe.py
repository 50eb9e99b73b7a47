"""Diagnostic: does running two time steps as a z-wavefront through the 256 MiB Infinity Cache pay?  (GPU needed)

Plain stepping streams the whole lattice through HBM once per step.  Here step t runs on a window of k lattice planes and
step t+1 follows on the planes that trail it by two (a cell of step t+1 needs every slot within +-2 planes to be finished
by step t: slots live in the cell's own plane and the two next to it, and their step-t owners one plane further), so the
DDFs that step t wrote are read back by step t+1 while they may still be cache-resident.  Every launch is an ordinary
lbmhip_stream_collide_planes call on one stream; the result is bit-identical to plain stepping (checked by CRC).

usage: python tools/fusion_probe.py [N] [storage]
"""
import ctypes as C
import sys
import zlib

import numpy as np

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L

lib = cfd_amd.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
storage = sys.argv[2] if len(sys.argv) > 2 else "FP32"
zero = C.c_float(0.0)


def make():
    cfg = L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, L.STORAGE[storage], 0, 0, 0.01, 0)
    dom = L.Domain(cfg)
    rng = np.random.RandomState(1)
    plane = (0.05 * rng.standard_normal((3, n * n))).astype(np.float32)
    for z in range(n):  # a non-trivial velocity field, uploaded plane by plane
        for c in range(3):
            dom.upload(L.FIELD_U, np.roll(plane[c], z * 17), offset=(c * dom.N + z * n * n) * 4)
    L.check(lib.lbmhip_initialize(dom.h))
    L.check(lib.lbmhip_set_tuning(dom.h, b"autotune", 0))
    return dom


def planes(dom, t, z0, z1):
    if z1 > z0:
        L.check(lib.lbmhip_stream_collide_planes(dom.h, C.c_uint64(t), zero, zero, zero, z0, z1))


def plain_pair(dom, t):
    planes(dom, t, 0, n)
    planes(dom, t + 1, 0, n)


def fused_pair(dom, t, k):
    """steps t and t+1; step t+1 trails step t by two planes; the planes next to the periodic seam come last"""
    done1 = 2  # step t+1 has finished planes [2, done1)
    for z in range(0, n, k):
        z1 = min(n, z + k)
        planes(dom, t, z, z1)
        upto = z1 - 2  # step t is finished on [0, z1): step t+1 may run below z1-2
        if z1 == n:
            upto = n - 2
        planes(dom, t + 1, done1, upto)
        done1 = max(done1, upto)
    planes(dom, t + 1, n - 2, n)
    planes(dom, t + 1, 0, 2)


def crc(dom):
    return zlib.crc32(dom.download_fi().tobytes())


def timed(dom, fn, pairs):
    ms = C.c_float()
    L.check(lib.lbmhip_timer_start(dom.h))
    t = 0
    for _ in range(pairs):
        fn(dom, t); t += 2
    L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
    return ms.value / (2 * pairs)


B = lib.lbmhip_bandwidth_bytes_per_cell(19, L.STORAGE[storage], 0)
ref = make()
try:
    for _ in range(3):
        plain_pair(ref, 0)
    per = timed(ref, plain_pair, 20)
    print("%s %d^3 plain            %.3f ms/step  %7.0f MLUPs/s  %.1f %% of 8 TB/s" % (storage, n, per, n ** 3 / per / 1e3, B * n ** 3 / per / 1e6 / 80), flush=True)
    want = None
    if n <= 256:
        chk = make(); plain_pair(chk, 0); want = crc(chk); chk.close()
    for k in (2, 3, 4, 6, 8, 16):
        dom = make()
        try:
            if want is not None:
                fused_pair(dom, 0, k)
                assert crc(dom) == want, "fused stepping differs from plain stepping (k=%d)" % k
            fn = lambda d, t: fused_pair(d, t, k)
            for _ in range(3):
                fn(dom, 0)
            per = timed(dom, fn, 20)
            print("%s %d^3 fused pairs k=%-3d %.3f ms/step  %7.0f MLUPs/s  %.1f %% of 8 TB/s%s" % (
                storage, n, k, per, n ** 3 / per / 1e3, B * n ** 3 / per / 1e6 / 80, "  (bit-identical)" if want is not None else ""), flush=True)
        finally:
            dom.close()
finally:
    ref.close()
