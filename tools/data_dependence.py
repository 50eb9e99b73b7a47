import ctypes as C, sys
sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import setups
from cfd_amd.lbm import check
L = cfd_amd.lib()
for storage in ("FP32", "FP16S", "FP16C"):
    for mode in ("tgv", "rest"):
        with cfd_amd.LBM((512, 512, 512), 1.0 if mode == "rest" else 0.01, storage=storage) as lbm:
            if mode == "tgv":
                setups.taylor_green(lbm)
            lbm.run(10)
            dom = lbm.domains[0]
            best = 1e9
            for _ in range(3):
                ms = C.c_float()
                check(L.lbmhip_timer_start(dom.h))
                for _ in range(10):
                    lbm.do_time_step()
                check(L.lbmhip_timer_stop(dom.h, C.byref(ms)))
                best = min(best, ms.value / 10)
            print(storage, mode, "%.3f ms/step %.0f MLUPs/s" % (best, lbm.get_N() / best / 1e3), flush=True)
