"""Diagnostic: rate of update_fields or initialize on a 512^3 D3Q19 lattice, in the output format of tools/config_bench.py so
that tools/profile_config.py can profile them (GPU needed).
usage: python tools/aux_bench.py <update_fields|initialize>_<fp32|fp16s|fp16c> [--steps=n] [--geometry=...] (the geometry is accepted and ignored)"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L

lib = cfd_amd.lib()
name = [a for a in sys.argv[1:] if not a.startswith("--")][0]
steps = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--steps=")), 20)
kernel, storage = name.rsplit("_", 1)
storage = storage.upper()
n = 512
fb = 4 if storage == "FP32" else 2
dom = L.Domain(L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, L.STORAGE[storage], 0, 0, 0.01, 0))
try:
    zero = C.c_float(0)
    L.check(lib.lbmhip_initialize(dom.h))
    if kernel == "update_fields":
        call = lambda: L.check(lib.lbmhip_update_fields(dom.h, C.c_uint64(0), zero, zero, zero))
    else:
        call = lambda: L.check(lib.lbmhip_initialize(dom.h))
    call(); call()
    ms = C.c_float()
    L.check(lib.lbmhip_timer_start(dom.h))
    for _ in range(steps):
        call()
    L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
    per = ms.value / steps
    B = 19 * fb + 17  # reads Q DDFs + flags, writes rho and u (update_fields); reads rho, u, flags, writes Q DDFs (initialize)
    print("%-26s %5dx%4dx%4d  %-22s %.3f ms/step  %7.0f MLUPs/s  %4d B/cell  %5.0f GB/s (%.1f%% of 8 TB/s)  [tuned=0 V=0 P=0 blocks/CU<=-1 block=0]" % (
        name, n, n, n, "default", per, dom.N / per / 1e3, B, B * dom.N / per / 1e6, B * dom.N / per / 1e6 / 80), flush=True)
finally:
    dom.close()
