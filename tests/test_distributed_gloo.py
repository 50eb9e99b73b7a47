"""world_size-2 (and 3) run of the multi-process halo exchange on CPU ranks (gloo backend).

The exchange logic under test is cfd_amd.distributed.DistributedLBM; the slab engine is the CPU oracle
(tests/slab_oracle.py), so the result must equal the oracle's single-process multi-domain run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, kw, setup_name, steps, overlap, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfd_amd import setups
        from cfd_amd.distributed import DistributedLBM
        from tests.slab_oracle import OracleSlabOps
        from cfd_amd import decomposition
        kw = dict(kw)
        N = kw.pop("N")
        local = decomposition.local_size(decomposition.rounded_size(N, (1, 1, world)), (1, 1, world))
        engine = lambda: OracleSlabOps(local, world, kw.get("Q", 19), kw.get("storage", "FP32"), kw.get("collision", "SRT"), kw.get("features", ()), kw["nu"])
        lbm = DistributedLBM(N, rank=rank, world=world, overlap=overlap, ops_factory=engine, **kw)
        getattr(setups, setup_name)(lbm)
        lbm.run(steps)
        lbm.read_from_device()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rho=lbm.rho, u=lbm.u, flags=lbm.flags)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, True), (2, False), (3, True)])
def test_slab_exchange_matches_single_process(tmp_path, world, overlap):
    sys.path.insert(0, ROOT)
    from cfd_amd import setups
    from oracle import harness as H
    kw = dict(N=(16, 12, 12 * world), nu=0.02, Q=19, storage="FP32", collision="SRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0))
    steps = 12
    mp.spawn(_worker, args=(world, _free_port(), kw, "channel", steps, overlap, str(tmp_path)), nprocs=world, join=True)
    ref = H.OracleLBM(kw["N"], kw["nu"], D=(1, 1, world), Q=19, features=kw["features"], f=kw["f"])
    setups.channel(ref)
    ref.run(steps)
    ref.read_from_device()
    nz = kw["N"][2] // world
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        sl = slice(r * nz, (r + 1) * nz)
        assert np.array_equal(got["flags"], ref.flags[sl])
        assert np.array_equal(got["rho"].view(np.uint32), ref.rho[sl].view(np.uint32))
        assert np.array_equal(got["u"].view(np.uint32), ref.u[:, sl].view(np.uint32))
