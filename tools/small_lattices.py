import ctypes as C, sys, time
sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import setups
from cfd_amd.lbm import check
L = cfd_amd.lib()
for n in (32, 64, 96, 128, 192, 256):
    with cfd_amd.LBM((n, n, n), 0.01) as lbm:
        setups.taylor_green(lbm)
        lbm.run(20)
        dom = lbm.domains[0]
        steps = 400
        ms = C.c_float()
        t0 = time.perf_counter()
        check(L.lbmhip_timer_start(dom.h))
        for _ in range(steps):
            lbm.do_time_step()
        check(L.lbmhip_timer_stop(dom.h, C.byref(ms)))
        wall = time.perf_counter() - t0
        print("%4d^3: GPU %.2f us/step, wall %.2f us/step, %.0f MLUPs/s (ideal at 5.4 TB/s: %.2f us)" % (n, ms.value / steps * 1e3, wall / steps * 1e6, n ** 3 / (wall / steps) / 1e6, n ** 3 * 153 / 5.4e12 * 1e6), flush=True)
