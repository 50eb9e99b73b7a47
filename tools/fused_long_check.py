"""Diagnostic: 1000 steps of the 512^3 headline lattice with two steps per launch against one launch per step, DDFs compared
bit for bit (CRC-32 per stream).  (GPU needed; about a minute)   usage: python tools/fused_long_check.py [storage] [steps]
With 16-bit DDFs use about 200 steps: later on this vortex (A = 0.25) leaves the FP16S range and every stream holds one value."""
import sys
import zlib

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfd_amd
from cfd_amd import setups

storage = sys.argv[1] if len(sys.argv) > 1 else "FP32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
crcs = {}
for fuse in (-1, 0):
    lbm = cfd_amd.LBM(N=(512, 512, 512), nu=0.01, storage=storage)
    lbm.set_tuning("fuse", fuse)
    setups.taylor_green(lbm)
    import ctypes as C
    cfd_amd.lbm.check(cfd_amd.lib().lbmhip_set_tuning(lbm.domains[0].h, b"autotune", 0))
    lbm.run(steps)
    v = C.c_int(0); cfd_amd.lbm.check(cfd_amd.lib().lbmhip_get_tuning(lbm.domains[0].h, b"fused", C.byref(v)))
    fi = lbm.domains[0].download_fi()
    n = 512 ** 3
    crcs[fuse] = [zlib.crc32(fi[i * n:(i + 1) * n].tobytes()) for i in range(19)]
    print("fuse %2d: form of the paired launches %d, launches that gave up waits %d, CRC of stream 0 %08x, of stream 18 %08x" % (fuse, v.value, lbm.get_tuning("band_given_up"), crcs[fuse][0], crcs[fuse][18]), flush=True)
    lbm.close()
print("%s, %d steps: %s" % (storage, steps, "bit-identical" if crcs[-1] == crcs[0] else "DIFFERENT"))
sys.exit(0 if crcs[-1] == crcs[0] else 1)
