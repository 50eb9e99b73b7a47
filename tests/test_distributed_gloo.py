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
        D = kw.get("D") or (1, 1, world)
        local = decomposition.local_size(decomposition.rounded_size(N, D), D)
        engine = lambda: OracleSlabOps(local, D, kw.get("Q", 19), kw.get("storage", "FP32"), kw.get("collision", "SRT"), kw.get("features", ()), kw["nu"])
        if setup_name == "channel_planes":  # no host copy of the fields: flags go into the slab plane by plane
            lbm = DistributedLBM(N, rank=rank, world=world, overlap=overlap, ops_factory=engine, host_fields=False, **kw)
            G = lbm.N
            lbm.fill_planes(lambda z: {"flags": setups.channel_flag_plane(G[0], G[1], G[2], z)})
            lbm.run(steps)
            lbm.update_fields()
            Lx, Ly, Lz = lbm.local
            got = lbm.ops.download(("rho", "u", "flags"))
            zs = slice(lbm.hz, Lz - lbm.hz)
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rho=got["rho"].reshape(Lz, Ly, Lx)[zs], u=got["u"].reshape(3, Lz, Ly, Lx)[:, zs],
                     flags=got["flags"].reshape(Lz, Ly, Lx)[zs])
            return
        lbm = DistributedLBM(N, rank=rank, world=world, overlap=overlap, ops_factory=engine, **kw)
        getattr(setups, setup_name)(lbm)
        lbm.run(steps)
        lbm.read_from_device()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rho=lbm.rho, u=lbm.u, flags=lbm.flags)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap,setup", [(2, True, "channel"), (2, False, "channel"), (3, True, "channel"), (2, True, "channel_planes")])
def test_slab_exchange_matches_single_process(tmp_path, world, overlap, setup):
    sys.path.insert(0, ROOT)
    from cfd_amd import setups
    from oracle import harness as H
    kw = dict(N=(16, 12, 12 * world), nu=0.02, Q=19, storage="FP32", collision="SRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0))
    steps = 15  # seven pairs of steps (every second exchange packed by one pair and finished beside the next) and a single one
    mp.spawn(_worker, args=(world, _free_port(), kw, setup, steps, overlap, str(tmp_path)), nprocs=world, join=True)
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


@pytest.mark.parametrize("D,setup", [((2, 2, 1), "channel"), ((2, 1, 2), "channel"), ((1, 2, 2), "taylor_green"), ((2, 2, 2), "channel"), ((2, 2, 1), "channel_planes_blocks")])
def test_block_exchange_matches_single_process(tmp_path, D, setup):
    """Dx x Dy x Dz blocks, one rank each (lbm.cpp:1038-1073): the exchange runs x, then y, then z, so that edge and corner cells
    arrive through the halos of the axes before (lbm.cpp:2162-2181).  Every rank's block equals its part of the undivided lattice."""
    sys.path.insert(0, ROOT)
    from cfd_amd import setups, decomposition
    from oracle import harness as H
    world = D[0] * D[1] * D[2]
    kw = dict(N=(12, 10, 8), nu=0.02, Q=19, storage="FP32", collision="SRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 2e-5, 0.0), D=D)
    steps = 7
    worker_setup = {"channel_planes_blocks": "channel_planes"}.get(setup, setup)
    mp.spawn(_worker, args=(world, _free_port(), kw, worker_setup, steps, True, str(tmp_path)), nprocs=world, join=True)
    ref = H.OracleLBM(kw["N"], kw["nu"], D=(1, 1, 1), Q=19, features=kw["features"], f=kw["f"])
    (setups.taylor_green if setup == "taylor_green" else setups.channel)(ref)
    ref.run(steps)
    ref.read_from_device()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        glob, loc = decomposition.slices(r, kw["N"], D)
        if setup == "channel_planes_blocks":  # saved with the halo cells of x and y, without those of z
            got = {k: got[k][..., loc[1], loc[2]] for k in ("rho", "u", "flags")}
        assert np.array_equal(got["flags"], ref.flags[glob])
        assert np.array_equal(got["rho"].view(np.uint32), ref.rho[glob].view(np.uint32))
        assert np.array_equal(got["u"].view(np.uint32), ref.u[(slice(None),) + glob].view(np.uint32))


# ---- which buffer meets which: the message pairing of the exchange for 2, 3 and 8 ranks, without any process group
class _FakeDist:
    """Stands in for torch.distributed inside cfd_amd.distributed: records the point-to-point operations every rank issues.
    deliver() then moves the data the way the two real backends match messages: RCCL pairs the k-th send from a to b with
    the k-th receive b posts for a (tags are ignored); gloo pairs by (source, tag)."""

    class P2POp:
        def __init__(self, op, tensor, peer, group=None, tag=0):
            self.op, self.tensor, self.peer, self.tag = op, tensor, peer, tag

    class _Done:
        def wait(self):
            pass

    isend, irecv = "isend", "irecv"

    def __init__(self):
        self.rank = None
        self.posted = []  # (rank, op, tensor, peer, tag) in issue order

    def get_backend(self, group=None):
        return "fake"

    def batch_isend_irecv(self, ops):
        for o in ops:
            self.posted.append((self.rank, o.op, o.tensor, o.peer, o.tag))
        return [self._Done() for _ in ops]

    def deliver(self, by_tag):
        sends = [p for p in self.posted if p[1] == "isend"]
        recvs = [p for p in self.posted if p[1] == "irecv"]
        used = set()
        for dst, _, rbuf, src, rtag in recvs:
            for k, (s_rank, _, sbuf, s_peer, stag) in enumerate(sends):
                if k in used or s_rank != src or s_peer != dst or (by_tag and stag != rtag):
                    continue
                used.add(k)
                assert sbuf.numel() == rbuf.numel()
                rbuf.copy_(sbuf)
                break
            else:
                raise AssertionError("receive of rank %d from %d (tag %d) has no matching send" % (dst, src, rtag))
        assert len(used) == len(sends), "a send was never received"


class _BufferOps:
    local = (4, 3, 6)

    def __init__(self, rank):
        # every byte names its origin: rank in the low bits, face in bit 6 (+ face) or bit 7 (- face)
        mk = lambda v: torch.full((64,), v, dtype=torch.uint8)
        self.buffers = {"send_p": mk(0x40 | rank), "send_m": mk(0x80 | rank), "recv_p": mk(0xFF), "recv_m": mk(0xFF)}


@pytest.mark.parametrize("by_tag", [False, True])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_exchange_pairs_plus_faces_with_minus_halos(monkeypatch, world, by_tag):
    """lbm.cpp:2129-2136, 2162-2181: the + face of slab d becomes the - halo of slab d+1 and the - face of slab d+1 the
    + halo of slab d, with periodic wrap; with two ranks both messages of a rank go to the same peer."""
    sys.path.insert(0, ROOT)
    import cfd_amd.distributed as DD
    fake = _FakeDist()
    monkeypatch.setattr(DD, "dist", fake)
    ranks = []
    for r in range(world):
        lbm = DD.DistributedLBM((4, 3, 4 * world), 0.02, rank=r, world=world, ops_factory=lambda r=r: _BufferOps(r), host_fields=False)
        ranks.append(lbm)
    for r, lbm in enumerate(ranks):
        fake.rank = r
        lbm._exchange(48)  # a message shorter than the buffers: only the first 48 bytes travel
    fake.deliver(by_tag)
    for r, lbm in enumerate(ranks):
        b = lbm.ops.buffers
        prev, nxt = (r - 1) % world, (r + 1) % world
        assert (b["recv_m"][:48] == (0x40 | prev)).all(), "the - halo must hold the + face of the previous slab"
        assert (b["recv_p"][:48] == (0x80 | nxt)).all(), "the + halo must hold the - face of the next slab"
        assert (b["recv_m"][48:] == 0xFF).all() and (b["recv_p"][48:] == 0xFF).all()
