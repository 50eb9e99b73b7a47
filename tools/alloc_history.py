"""Diagnostic: does the stream_collide rate of a 512^3 FP32 lattice depend on what was allocated and freed before it?
(physical placement of the fields decides how much address space one TLB entry covers)  GPU needed."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L

lib = cfd_amd.lib()


def rate(N, storage="FP32", label=""):
    cfg = L.Config(N[0], N[1], N[2], 1, 1, 1, 0, 0, 0, 19, L.STORAGE[storage], 0, 0, 0.01, 0)
    dom = L.Domain(cfg)
    try:
        L.check(lib.lbmhip_set_tuning(dom.h, b"autotune", 0))
        L.check(lib.lbmhip_initialize(dom.h))
        t = 0
        best = 1e9
        for rep in range(3):
            for _ in range(5):
                L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(t), C.c_float(0), C.c_float(0), C.c_float(0))); t += 1
            ms = C.c_float()
            L.check(lib.lbmhip_timer_start(dom.h))
            for _ in range(20):
                L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(t), C.c_float(0), C.c_float(0), C.c_float(0))); t += 1
            L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
            best = min(best, ms.value / 20)
        p, n = dom.field_pointer(L.FIELD_FI)
        print("%-44s %4dx%4dx%4d %s  %.3f ms  %.0f MLUPs/s  fi at 0x%x" % (label, N[0], N[1], N[2], storage, best, dom.N / best / 1e3, p), flush=True)
    finally:
        dom.close()


rate((512, 512, 512), label="fresh process")
rate((512, 512, 512), label="again")
rate((1024, 1024, 1024), label="big lattice (88 GB)")
rate((512, 512, 512), label="after the big lattice was freed")
held = [L.Domain(L.Config(256, 256, 256, 1, 1, 1, 0, 0, 0, 19, 0, 0, 0, 0.01, 0)) for _ in range(8)]
rate((512, 512, 512), label="with 8 small lattices alive")
for d in held[::2]:
    d.close()
rate((512, 512, 512), label="after freeing every other small lattice")
for d in held[1::2]:
    d.close()
rate((1600, 512, 512), label="1600x512x512")
rate((512, 512, 512), label="last")
