"""Run as a script by tests/test_distributed_gpu.py (a fresh process, so that the library resolves the RCCL entry points to
tests/native/libfake_rccl.so): W slabs on one GPU exchange their halos through lbmhip_halo_exchange_rccl, each slab as rank k
of a W-rank job, and must end up bit-identical to the same slabs exchanged by lbmhip_halo_exchange_local."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LBMHIP_RCCL_LIB"] = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")
fake = C.CDLL(os.environ["LBMHIP_RCCL_LIB"])

import cfd_amd  # noqa: E402
from cfd_amd import setups  # noqa: E402
from cfd_amd.lbm import check  # noqa: E402


class RankwiseLBM(cfd_amd.LBM):
    """Every domain plays one rank: pack, the four grouped send/recv calls of the C ABI, delivery, unpack."""
    messages = 0

    def _communicate(self, what):
        L = self._L
        W = len(self.domains)
        ranks = [C.c_int(k) for k in range(W)]
        for dom in self.domains:
            check(L.lbmhip_halo_pack(dom.h, what, 2, C.c_uint64(self.t)))
        for k, dom in enumerate(self.domains):
            check(L.lbmhip_halo_exchange_rccl(dom.h, C.byref(ranks[k]), what, 2, (k - 1) % W, (k + 1) % W))
        n = fake.fake_rccl_deliver()
        assert n == 2 * W, "fake RCCL delivery failed or miscounted: %d" % n
        RankwiseLBM.messages += n
        for dom in self.domains:
            check(L.lbmhip_halo_unpack(dom.h, what, 2, C.c_uint64(self.t)))


def run(cls, W, Q, storage, features):
    kw = dict(N=(24, 20, 8 * W), nu=0.02, D=(1, 1, W), Q=Q, storage=storage, features=features, f=(1e-4, 0.0, 0.0), devices=[0] * W, overlap=False)
    with cls(**kw) as lbm:
        setups.channel(lbm)
        if "FORCE_FIELD" in features:
            lbm.F[...] = 0.0
        lbm.run(9)
        lbm.read_from_device()
        return lbm.rho.copy(), lbm.u.copy(), lbm.flags.copy(), [d.download_fi().copy() for d in lbm.domains]


for W in (2, 3, 4):
    for Q, storage, features in ((19, "FP32", ("VOLUME_FORCE", "SUBGRID")), (27, "FP16C", ("VOLUME_FORCE", "FORCE_FIELD"))):
        want = run(cfd_amd.LBM, W, Q, storage, features)
        got = run(RankwiseLBM, W, Q, storage, features)
        assert np.array_equal(want[2], got[2])
        assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32)) and np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))
        for a, b in zip(want[3], got[3]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
        assert np.abs(want[1]).max() > 0
assert RankwiseLBM.messages > 0
print("rccl call order ok: %d messages" % RankwiseLBM.messages)
