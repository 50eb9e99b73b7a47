import sys, time
sys.path.insert(0, ".")
import cfd_amd
from tests import cases as CS
from tests.cases import CASES
for name in ("all_fp16s_trt_2dom", "all_fp32_srt_8dom", "q27_fp32_trt_4slab", "tgv32_fp32_srt"):
    case = CASES[name]
    t0 = time.time()
    with cfd_amd.LBM(**case["kw"]) as a:
        case["setup"](a)
        a.run(5)
        print(name, "jit =", a.get_tuning("jit"), "first 5 steps %.2f s" % (time.time() - t0), flush=True)
