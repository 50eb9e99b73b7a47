import ctypes as C, os, sys
sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import lbm as L
lib = cfd_amd.lib()
Nz = int(sys.argv[1]) if len(sys.argv) > 1 else 514
st = sys.argv[2] if len(sys.argv) > 2 else "FP32"
plane = 512 * 512
pads = [0, 256, 2048, 16384, 32768, 65536, 65536 + 256, 98304, 131072, 196608, 1 << 20] if len(sys.argv) < 4 else [int(v) for v in sys.argv[3].split(',')]
for pad in pads:
    os.environ["LBMHIP_PAD"] = str(pad)
    cfg = L.Config(512, 512, Nz, 1, 1, 1, 0, 0, 0, 19, L.STORAGE[st], 0, 0, 0.01, 0)
    dom = L.Domain(cfg)
    L.check(lib.lbmhip_initialize(dom.h))
    step = lambda t: L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(t), C.c_float(0), C.c_float(0), C.c_float(0)))
    for t in range(100): step(t)
    ms = C.c_float()
    L.check(lib.lbmhip_timer_start(dom.h))
    for t in range(100, 160): step(t)
    L.check(lib.lbmhip_timer_stop(dom.h, C.byref(ms)))
    per = ms.value / 60
    N = 512 * 512 * Nz
    print("Nz=%d %s pad=%9d elements (stride %.3f MiB): %.3f ms  %.0f MLUPs/s" % (Nz, st, pad, (N + pad) * (4 if st == "FP32" else 2) / 2**20, per, N / per / 1e3), flush=True)
    dom.close()
