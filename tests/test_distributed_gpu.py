"""GPU: the one-process-per-GPU runner (cfd_amd.distributed) with the HIP slab engine.  Two ranks share the one GPU of
the test box and exchange halos over gloo (host-staged rehearsal of the RCCL path); the result must equal the oracle's
two-domain run bit for bit.  Also: the largest single-GPU lattice of the benchmark configurations."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, kw, steps, overlap, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfd_amd import setups
        from cfd_amd.distributed import DistributedLBM
        kw = dict(kw)
        lbm = DistributedLBM(kw.pop("N"), rank=rank, world=world, overlap=overlap, **kw)
        setups.channel(lbm)
        lbm.run(steps)
        lbm.read_from_device()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rho=lbm.rho, u=lbm.u, flags=lbm.flags)
        lbm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_on_one_gpu_match_the_oracle(gpu, tmp_path, overlap):
    sys.path.insert(0, ROOT)
    from cfd_amd import setups
    from oracle import harness as H
    world, steps = 2, 20
    kw = dict(N=(64, 32, 48), nu=0.01, Q=19, storage="FP32", collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0))
    mp.spawn(_worker, args=(world, _free_port(), kw, steps, overlap, str(tmp_path)), nprocs=world, join=True)
    ref = H.OracleLBM(kw["N"], kw["nu"], D=(1, 1, world), Q=19, collision="TRT", features=kw["features"], f=kw["f"])
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


@pytest.mark.parametrize("D,storage", [((2, 2, 1), "FP32"), ((2, 1, 2), "FP16S"), ((2, 2, 2), "FP32")])
def test_blocks_on_one_gpu_match_the_oracle(gpu, tmp_path, D, storage):
    """Dx x Dy x Dz blocks, one rank each, all on this GPU (gloo, staged through the host): the HIP halo kernels of all three axes
    under the multi-process exchange, x then y then z (lbm.cpp:2162-2181).  Every block equals its part of the oracle's lattice."""
    sys.path.insert(0, ROOT)
    from cfd_amd import setups, decomposition
    from oracle import harness as H
    world, steps = D[0] * D[1] * D[2], 9
    kw = dict(N=(64, 24, 16), nu=0.01, Q=19, storage=storage, collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 3e-5, 0.0), D=D)
    mp.spawn(_worker, args=(world, _free_port(), kw, steps, True, str(tmp_path)), nprocs=world, join=True)
    ref = H.OracleLBM(kw["N"], kw["nu"], D=(1, 1, 1), Q=19, storage=storage, collision="TRT", features=kw["features"], f=kw["f"])
    setups.channel(ref)
    ref.run(steps)
    ref.read_from_device()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        glob, _ = decomposition.slices(r, kw["N"], D)
        assert np.array_equal(got["flags"], ref.flags[glob])
        assert np.array_equal(got["rho"].view(np.uint32), ref.rho[glob].view(np.uint32))
        assert np.array_equal(got["u"].view(np.uint32), ref.u[(slice(None),) + glob].view(np.uint32))


def _self_exchange_worker(rank, port, kw, steps, overlap, out_dir):
    """One rank, backend nccl (= RCCL): the rank is its own z-neighbour on both sides, so its slab with halo planes is a
    periodic lattice of nz planes.  Runs the production exchange code (device buffers, RCCL send/recv, streams)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from cfd_amd import setups
        from cfd_amd.distributed import DistributedLBM
        import cfd_amd
        kw = dict(kw)
        N = kw.pop("N")
        tuning = kw.pop("tuning", {})
        setup = getattr(setups, kw.pop("setup", "channel"))
        lbm = DistributedLBM((N[0], N[1], 2 * N[2]), rank=0, world=2, overlap=overlap, **kw)  # 2 slabs of N[2] planes; this rank is slab 0
        lbm.next = lbm.prev = 0
        for key, value in tuning.items():
            cfd_amd.lbm.check(cfd_amd.lib().lbmhip_set_tuning(lbm.ops.dom.h, key.encode(), int(value)))
        with cfd_amd.LBM(N, **kw) as single:
            single.set_tuning("fuse", 0)
            setup(single)
            lbm.rho[...] = single.rho; lbm.u[...] = single.u; lbm.flags[...] = single.flags
            single.run(steps)
            single.read_from_device()
            lbm.run(steps)
            import ctypes
            fused = ctypes.c_int(-1)  # of the last stream_collide call of the run
            cfd_amd.lbm.check(cfd_amd.lib().lbmhip_get_tuning(lbm.ops.dom.h, b"fused", ctypes.byref(fused)))
            lbm.read_from_device()
            np.savez(os.path.join(out_dir, "self.npz"), rho=lbm.rho, u=lbm.u, flags=lbm.flags, rho0=single.rho, u0=single.u, flags0=single.flags, fused=fused.value)
        lbm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_exchange_with_itself_equals_periodic_lattice(gpu, tmp_path, overlap):
    kw = dict(N=(64, 32, 24), nu=0.01, Q=19, storage="FP16S", collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0))
    mp.spawn(_self_exchange_worker, args=(_free_port(), kw, 20, overlap, str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(str(tmp_path), "self.npz"))
    assert np.array_equal(got["flags"], got["flags0"])
    assert np.array_equal(got["rho"].view(np.uint32), got["rho0"].view(np.uint32))
    assert np.array_equal(got["u"].view(np.uint32), got["u0"].view(np.uint32))
    assert float(np.abs(got["u"]).max()) > 1e-4


@pytest.mark.parametrize("storage,steps", [("FP32", 20), ("FP16S", 21)])
def test_slab_steps_paired_in_one_launch_equal_periodic_lattice(gpu, tmp_path, storage, steps):
    """The slab schedule that takes two steps at a time, with the inner planes of both steps in ONE launch (plain D3Q19 SRT has
    such a kernel; fuse = 1 uses it on this small lattice), over the production RCCL exchange of a rank that is its own neighbour."""
    kw = dict(N=(256, 32, 40), nu=0.01, Q=19, storage=storage, collision="SRT", features=(), setup="taylor_green", tuning={"fuse": 1, "pair_lag": 3})
    mp.spawn(_self_exchange_worker, args=(_free_port(), kw, steps, True, str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(str(tmp_path), "self.npz"))
    assert int(got["fused"]) == 1  # the inner planes did run two steps per launch (the odd run ends with a single step)
    assert np.array_equal(got["flags"], got["flags0"])
    assert np.array_equal(got["rho"].view(np.uint32), got["rho0"].view(np.uint32))
    assert np.array_equal(got["u"].view(np.uint32), got["u0"].view(np.uint32))
    assert float(np.abs(got["u"]).max()) > 1e-3


def test_capacity_lattice_1600_fp16s(gpu):
    """D3Q19 FP16S at 1600^3 (4.1e9 cells, 225 GB: the capacity configuration of the benchmark list).  Only a small
    block of cells is disturbed, so for a few steps its neighbourhood must evolve exactly like the same block in a small
    periodic box computed by the oracle - a check of 64-bit indexing at the far end of a 155 GB allocation."""
    import cfd_amd
    from cfd_amd import lbm as L
    from oracle import harness as H
    free, total = C.c_uint64(), C.c_uint64()
    L.check(cfd_amd.lib().lbmhip_device_memory(0, C.byref(free), C.byref(total)))
    n = 1600
    need = n ** 3 * cfd_amd.lib().lbmhip_bytes_per_cell_device(19, 1, 0)
    if free.value < need * 1.01:
        n = int((free.value * 0.95 / 55) ** (1.0 / 3.0)) // 64 * 64  # whatever fits (reported in the assertion messages)
    assert n >= 1024, "less than 62 GB of device memory free: %d" % free.value
    small, steps, blob = 48, 6, 8
    cfg = L.Config(n, n, n, 1, 1, 1, 0, 0, 0, 19, L.STORAGE["FP16S"], 0, 0, 0.02, 0)
    dom = L.Domain(cfg)
    try:
        lib = cfd_amd.lib()
        N = n ** 3
        assert N * 19 > 2 ** 32  # DDF element offsets need 64 bits
        # velocity blob near the END of the lattice: rows of `blob` cells at (x0.., y, z) with x0,y,z around n-24
        base = n - small // 2 - blob // 2
        rng = np.random.RandomState(3)
        ux = (rng.rand(blob, blob, blob).astype(np.float32) - 0.5) * 0.1
        for dz in range(blob):
            for dy in range(blob):
                off = ((base + dz) * n + (base + dy)) * n + base
                dom.upload(L.FIELD_U, np.ascontiguousarray(ux[dz, dy]), offset=off * 4)
        L.check(lib.lbmhip_initialize(dom.h))
        for t in range(steps):
            L.check(lib.lbmhip_stream_collide(dom.h, C.c_uint64(t), C.c_float(0), C.c_float(0), C.c_float(0)))
        L.check(lib.lbmhip_update_fields(dom.h, C.c_uint64(steps), C.c_float(0), C.c_float(0), C.c_float(0)))
        L.check(lib.lbmhip_sync(dom.h))
        # read back the small box around the blob, row by row
        o = n - small
        got = np.empty((3, small, small, small), dtype=np.float32)
        row = np.empty(small, dtype=np.float32)
        for c in range(3):
            for z in range(small):
                for y in range(small):
                    dom.download(L.FIELD_U, row, offset=(c * N + ((o + z) * n + (o + y)) * n + o) * 4)
                    got[c, z, y] = row
    finally:
        dom.close()
    ref = H.OracleLBM((small, small, small), 0.02, storage="FP16S")
    b0 = small // 2 - blob // 2
    ref.u[0, b0:b0 + blob, b0:b0 + blob, b0:b0 + blob] = ux
    ref.run(steps)
    ref.read_from_device()
    assert np.abs(got).max() > 1e-3
    assert np.array_equal(got.view(np.uint32), ref.u.view(np.uint32)), "n=%d" % n


def _rccl_abi_worker(_index, kw, steps, overlap, out_dir):
    """One rank whose slab is its own z-neighbour, halos exchanged by lbmhip_halo_exchange_rccl on a communicator made
    with RCCL's C API (no torch.distributed anywhere)."""
    sys.path.insert(0, ROOT)
    import torch
    torch.cuda.set_device(0)
    import cfd_amd
    from cfd_amd import setups
    from cfd_amd import distributed as DD
    from cfd_amd.lbm import check
    rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    lib = cfd_amd.lib()
    kw = dict(kw)
    N = kw.pop("N")
    lbm = DD.DistributedLBM((N[0], N[1], 2 * N[2]), rank=0, world=2, overlap=overlap, **kw)

    def communicate(what):
        lbm.ops.pack(what, lbm.t)
        check(lib.lbmhip_halo_exchange_rccl(lbm.ops.dom.h, comm, DD.WHAT[what], 2, 0, 0))
        lbm.ops.unpack(what, lbm.t)
    lbm._communicate = communicate
    with cfd_amd.LBM(N, **kw) as single:
        setups.channel(single)
        lbm.rho[...] = single.rho; lbm.u[...] = single.u; lbm.flags[...] = single.flags
        single.run(steps)
        single.read_from_device()
        lbm.run(steps)
        lbm.read_from_device()
        np.savez(os.path.join(out_dir, "abi.npz"), rho=lbm.rho, u=lbm.u, flags=lbm.flags, rho0=single.rho, u0=single.u, flags0=single.flags)
    # misuse is reported, not executed
    assert lib.lbmhip_halo_exchange_rccl(lbm.ops.dom.h, None, DD.WHAT["fi"], 2, 0, 0) != 0
    assert lib.lbmhip_halo_exchange_rccl(lbm.ops.dom.h, comm, DD.WHAT["fi"], 0, 0, 0) != 0  # x has a single domain
    lbm.ops.sync()
    lbm.close()
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_entry_point_of_the_abi(gpu, tmp_path, overlap):
    kw = dict(N=(64, 32, 24), nu=0.01, Q=27, storage="FP16C", collision="SRT", features=("VOLUME_FORCE",), f=(1e-4, 0.0, 0.0))
    mp.spawn(_rccl_abi_worker, args=(kw, 20, overlap, str(tmp_path)), nprocs=1, join=True)
    got = np.load(os.path.join(str(tmp_path), "abi.npz"))
    assert np.array_equal(got["flags"], got["flags0"])
    assert np.array_equal(got["rho"].view(np.uint32), got["rho0"].view(np.uint32))
    assert np.array_equal(got["u"].view(np.uint32), got["u0"].view(np.uint32))
    assert float(np.abs(got["u"]).max()) > 1e-4


def _script(args, timeout=600):
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, "%s failed:\n%s\n%s" % (" ".join(args), r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_rccl_call_order_pairs_faces_and_halos_for_2_3_4_ranks(gpu):
    """lbmhip_halo_exchange_rccl for several ranks with distinct peers (and with one peer on both sides, world 2): the slabs
    live on this GPU, the RCCL entry points are stand-ins that deliver messages in RCCL's matching order."""
    out = _script([os.path.join(ROOT, "tests", "rccl_order_check.py")])
    assert "rccl call order ok" in out


@pytest.mark.parametrize("extra", [[], ["--config", "channel"], ["--config", "q27_force", "--lattice", "96,64,64"]])
def test_bench_launches_its_own_ranks(gpu, extra):
    """`python bench.py --gpus 2` started plainly (no torch.distributed.run around it) runs two ranks and prints one JSON line.
    Both ranks share this GPU, so the transport is gloo; the driver's multi-GPU runs use the default, RCCL."""
    import json
    out = _script([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "64", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + extra)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and len(r["per_rank_mlups"]) == 2
    assert r["config"]["decomposition"] == "z-slabs x2" and r["config"]["launch_geometry"]["cells_per_thread"] >= 0
    assert r["roofline"]["frac"] > 0
    ranks = r["config"]["ranks"]  # where every rank sat
    assert len(ranks) == 2 and sorted(k["rank"] for k in ranks) == [0, 1] and all(k["device"] == 0 and k["pci_bus_id"] for k in ranks)
    assert ("fp16s" in r) == (not extra)  # the default configuration is measured with FP16S DDFs as well
    if not extra:
        assert r["fp16s"]["value"] > 0 and r["fp16s"]["roofline"]["frac"] > 0 and len(r["fp16s"]["per_rank_mlups"]) == 2
