"""GPU: lbmhip_stream_collide_steps (two time steps per launch on a single periodic domain) gives the bits of stepwise
lbmhip_stream_collide calls and of the CPU oracle.

The second step of a launch only waits for workgroups of the first; a short lag makes those waits real (workgroups of the
planes asked for are still running), a long one makes them trivial - both must give the same lattice.
"""
import ctypes as C

import numpy as np
import pytest

import cfd_amd
from cfd_amd import setups
from oracle import harness as H

pytestmark = pytest.mark.gpu

TGV = dict(N=(96, 64, 48), nu=0.02)
CHANNEL = dict(N=(320, 64, 40), nu=0.01, features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0), collision="TRT")
Q27 = dict(N=(64, 48, 40), nu=0.03, Q=27)
ODD_ROWS = dict(N=(66, 64, 48), nu=0.02)                    # rows that do not start on 16-byte boundaries: no packed stores
HALF_S = dict(N=(192, 64, 40), nu=0.02, storage="FP16S")   # 16-bit DDFs: two cells per lane, rows of several waves (seam lanes)
HALF_C = dict(N=(136, 48, 40), nu=0.02, storage="FP16C")   # a row that ends inside a wave


def fused_flag(lbm):
    v = C.c_int(-1)
    cfd_amd.lbm.check(cfd_amd.lib().lbmhip_get_tuning(lbm.domains[0].h, b"fused", C.byref(v)))
    return v.value


def tuning(lbm, key):
    v = C.c_int(-1)
    cfd_amd.lbm.check(cfd_amd.lib().lbmhip_get_tuning(lbm.domains[0].h, key.encode(), C.byref(v)))
    return v.value


def bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint16)


@pytest.mark.parametrize("lag,cap", [(3, 0), (5, 0), (4, 3), (9, 2)])
@pytest.mark.parametrize("kw,setup,steps,together", [(TGV, setups.taylor_green, 7, 1), (ODD_ROWS, setups.taylor_green, 4, 1), (CHANNEL, setups.channel, 10, 1), (Q27, setups.taylor_green, 4, 1),
                                                     (HALF_S, setups.taylor_green, 6, 1), (HALF_S, setups.taylor_green, 5, 2), (HALF_C, setups.taylor_green, 6, 1)])
def test_paired_steps_equal_single_steps(gpu, kw, setup, steps, together, lag, cap):
    a = cfd_amd.LBM(**kw); b = cfd_amd.LBM(**kw)
    try:
        for lbm, fuse in ((a, 1), (b, 0)):
            lbm.set_tuning("fuse", fuse)
            setup(lbm)
        a.set_tuning("pair_lag", lag); a.set_tuning("pair_blocks_per_cu", cap); a.set_tuning("pair_cells_together", together)
        a.run(steps); b.run(steps)
        assert fused_flag(a) == 1 and fused_flag(b) == 0
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi())), "DDFs differ"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
        a.run(1); b.run(1)  # a single step after paired ones
        assert fused_flag(a) == 0
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi())), "DDFs differ after the odd step"
    finally:
        a.close(); b.close()


BAND_TGV = dict(N=(96, 72, 24), nu=0.02)                     # blocks of 9 rows
BAND_RAGGED = dict(N=(320, 45, 20), nu=0.01, features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0), collision="TRT")  # blocks of 6 rows, the last one of 3; rows of two chunks


@pytest.mark.parametrize("rows,lag,cap", [(16, 2, 0), (3, 2, 0), (4, 5, 2), (2, 3, 4), (64, 4, 3)])
@pytest.mark.parametrize("kw,setup,steps,together", [(TGV, setups.taylor_green, 7, 1), (BAND_TGV, setups.taylor_green, 6, 1), (BAND_RAGGED, setups.channel, 6, 1), (CHANNEL, setups.channel, 6, 1),
                                                     (Q27, setups.taylor_green, 4, 1), (HALF_S, setups.taylor_green, 6, 1), (HALF_S, setups.taylor_green, 5, 2), (HALF_C, setups.taylor_green, 6, 1)])
def test_banded_steps_equal_single_steps(gpu, kw, setup, steps, together, rows, lag, cap):
    """Two steps per launch band by band on each XCD (fuse = 2): every band height (two rows, bands that do not divide the block, a
    band larger than the block), short and long lags, with and without a residency cap - always the bits of single launches."""
    a = cfd_amd.LBM(**kw); b = cfd_amd.LBM(**kw)
    try:
        for lbm, fuse in ((a, 2), (b, 0)):
            lbm.set_tuning("fuse", fuse)
            setup(lbm)
        a.set_tuning("band_rows", rows); a.set_tuning("band_lag", lag); a.set_tuning("band_blocks_per_cu", cap); a.set_tuning("pair_cells_together", together)
        a.run(steps); b.run(steps)
        assert fused_flag(a) == 2 and fused_flag(b) == 0
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi())), "DDFs differ"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
        assert tuning(a, "band_given_up") == 0
        a.run(1); b.run(1)  # a single step after paired ones
        assert fused_flag(a) == 0
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi())), "DDFs differ after the odd step"
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("rows,lag", [(16, 2), (3, 2), (2, 3), (4, 5)])
@pytest.mark.parametrize("per_launch", [3, 4])
@pytest.mark.parametrize("kw,setup,steps", [(TGV, setups.taylor_green, 9), (BAND_TGV, setups.taylor_green, 7), (BAND_RAGGED, setups.channel, 8), (HALF_S, setups.taylor_green, 7), (Q27, setups.taylor_green, 6)])
def test_three_and_four_steps_per_launch_equal_single_steps(gpu, kw, setup, steps, per_launch, rows, lag):
    """More than two steps in one banded launch (every later step one row further down, one row more left at the block edges to the
    launches behind), followed by a shorter launch or a single step for the rest: the bits of single launches."""
    a = cfd_amd.LBM(**kw); b = cfd_amd.LBM(**kw)
    try:
        for lbm, fuse in ((a, 2), (b, 0)):
            lbm.set_tuning("fuse", fuse)
            setup(lbm)
        a.set_tuning("band_rows", rows); a.set_tuning("band_lag", lag); a.set_tuning("band_steps", per_launch)
        a.run(steps); b.run(steps)
        blocks_of = (kw["N"][1]+7)//8
        assert fused_flag(a) == 2 and tuning(a, "band_steps") == min(per_launch, blocks_of//2)
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi())), "DDFs differ"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
        assert tuning(a, "band_given_up") == 0
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("fuse", [1, 2])
def test_waits_that_run_out_are_redone(gpu, fuse):
    """A wait that gives up at once (no second poll allowed, the shortest lag: the first step's workgroups are still running) leaves
    its task to the launch behind; the lattice is still that of single launches, the library reports it and stops pairing.
    Both forms: k_collide_pair and k_collide_band."""
    kw = dict(N=(128, 64, 32), nu=0.02)
    a = cfd_amd.LBM(**kw); b = cfd_amd.LBM(**kw)
    try:
        a.set_tuning("fuse", fuse); b.set_tuning("fuse", 0)
        a.set_tuning("pair_lag", 3); a.set_tuning("band_rows", 4); a.set_tuning("band_lag", 2); a.set_tuning("band_spin_limit", 0)
        setups.taylor_green(a); setups.taylor_green(b)
        a.run(6); b.run(6)
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi())), "DDFs differ"
        assert tuning(a, "band_given_up") >= 1 and tuning(a, "fuse") == 0
        a.run(4); b.run(4)
        assert fused_flag(a) == 0
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi()))
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("kw,setup", [(CHANNEL, setups.channel), (HALF_S, setups.taylor_green), (BAND_RAGGED, setups.channel)])
def test_banded_steps_equal_the_oracle(gpu, kw, setup):
    """Bands of two rows: three and more bands per XCD block, also on the lattice of 45 rows (blocks of 6 rows, the last one of 3)."""
    a = cfd_amd.LBM(**kw); b = H.OracleLBM(**kw)
    try:
        a.set_tuning("fuse", 2); a.set_tuning("band_rows", 2); a.set_tuning("band_lag", 2)
        setup(a); setup(b)
        a.run(12); b.run(12)
        assert fused_flag(a) == 2
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].fi)), "DDFs differ from the oracle"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(a.flags, b.flags)
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
    finally:
        a.close()


@pytest.mark.parametrize("storage", ["FP32", "FP16S"])
def test_automatic_pairing_equals_the_oracle(gpu, storage):
    """Nothing set by hand: planes of 512 x 512 cells pair automatically, with the automatic band height and lag, and give the oracle's bits."""
    kw = dict(N=(512, 512, 12), nu=0.01, storage=storage)
    a = cfd_amd.LBM(**kw); b = H.OracleLBM(**kw)
    try:
        a.set_tuning("autotune", 0)
        setups.taylor_green(a); setups.taylor_green(b)
        a.run(6); b.run(6, threads=16)
        assert fused_flag(a) == 2 and tuning(a, "fuse") == -1 and tuning(a, "band_given_up") == 0
        assert tuning(a, "xcd_round_robin") == 1  # the placement probe at domain creation agrees with what the launches found
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].fi)), "DDFs differ from the oracle"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
    finally:
        a.close()


@pytest.mark.parametrize("kw,setup,pairs", [
    (dict(N=(256, 256, 12), nu=0.01, Q=27), setups.taylor_green, True),
    (dict(N=(256, 256, 12), nu=0.01, collision="TRT"), setups.taylor_green, True),
    (dict(N=(256, 256, 12), nu=0.01, features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setups.inflow_box, True),
    (dict(N=(256, 256, 12), nu=0.01, features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 1e-5, 0.0)), setups.channel, True),
    (dict(N=(320, 264, 12), nu=0.01, Q=27, storage="FP16C"), setups.taylor_green, True),
    (dict(N=(256, 256, 12), nu=0.01, features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0), collision="TRT"), setups.channel, False),   # has an instance, measured slower
    (dict(N=(256, 256, 12), nu=0.01, features=("FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES")), setups.sphere_force, True),
    (dict(N=(256, 256, 12), nu=0.01, features=("MOVING_BOUNDARIES",)), setups.taylor_green, False)])                                  # has none
def test_automatic_pairing_by_feature_set(gpu, kw, setup, pairs):
    """Automatic mode goes by what was measured per compiled instance: the sets that gain pair (and give the oracle's bits), the others
    keep a launch per step."""
    a = cfd_amd.LBM(**kw); b = H.OracleLBM(**kw)
    try:
        a.set_tuning("autotune", 0)
        setup(a); setup(b)
        a.run(6); b.run(6, threads=16)
        assert fused_flag(a) == (2 if pairs else 0) and tuning(a, "fuse") == -1 and tuning(a, "band_given_up") == 0
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].fi)), "DDFs differ from the oracle"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
    finally:
        a.close()


@pytest.mark.parametrize("kw,setup", [(CHANNEL, setups.channel), (HALF_S, setups.taylor_green), (HALF_C, setups.taylor_green)])
def test_paired_steps_equal_the_oracle(gpu, kw, setup):
    a = cfd_amd.LBM(**kw); b = H.OracleLBM(**kw)
    try:
        a.set_tuning("fuse", 1); a.set_tuning("pair_lag", 3)
        setup(a); setup(b)
        a.run(12); b.run(12)
        assert fused_flag(a) == 1
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].fi)), "DDFs differ from the oracle"
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(a.flags, b.flags)
        assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
    finally:
        a.close()


def test_what_is_never_paired(gpu):
    """UPDATE_FIELDS, several domains, lattices with too few planes and feature sets without an instance run one launch per step."""
    for kw in (dict(N=(64, 64, 48), nu=0.02, features=("UPDATE_FIELDS",)), dict(N=(64, 64, 48), nu=0.02, D=(1, 1, 2)), dict(N=(64, 64, 6), nu=0.02),
               dict(N=(64, 64, 48), nu=0.02, features=("MOVING_BOUNDARIES",)), dict(N=(64, 64, 48), nu=0.02, storage="FP16S", collision="TRT"), dict(N=(65, 64, 48), nu=0.02, storage="FP16S")):
        a = cfd_amd.LBM(**kw); b = H.OracleLBM(**kw)
        try:
            a.set_tuning("fuse", 1)
            setups.taylor_green(a); setups.taylor_green(b)
            a.run(4); b.run(4)
            assert fused_flag(a) == 0, kw
            for da, db in zip(a.domains, b.domains):
                assert np.array_equal(bits(da.download_fi()), bits(db.fi)), kw
        finally:
            a.close()


def test_large_planes_pair_band_by_band(gpu):
    """Planes far larger than any cache (160 MB) pair automatically - band by band, the lines between the two steps are those of a
    band - and k_collide_pair, whose second step reads whole planes back, runs on request only; both leave the bits of single launches."""
    kw = dict(N=(2048, 1024, 16), nu=0.02)
    a = cfd_amd.LBM(**kw); b = cfd_amd.LBM(**kw)
    try:
        b.set_tuning("fuse", 0)
        setups.taylor_green(a); setups.taylor_green(b)
        a.run(2); b.run(2)
        assert fused_flag(a) == 2 and fused_flag(b) == 0
        a.set_tuning("fuse", 1)
        a.run(2); b.run(2)
        assert fused_flag(a) == 1
        assert np.array_equal(bits(a.domains[0].download_fi()), bits(b.domains[0].download_fi()))
    finally:
        a.close(); b.close()


def test_pair_planes_entry_point(gpu):
    """lbmhip_stream_collide_pair_planes(t, [z0, z1)) = step t on [z0, z1) followed by step t+1 on [z0+2, z1-2), whether it runs as one
    launch or as two; plane ranges that reach into a halo are refused."""
    L = cfd_amd.lib()
    zero = C.c_float(0.0)
    kw = dict(N=(128, 32, 40), nu=0.02)
    got = {}
    for name, fuse in (("one launch", 1), ("two launches", 0), ("by hand", 0)):
        with cfd_amd.LBM(**kw) as lbm:
            lbm.set_tuning("fuse", fuse); lbm.set_tuning("pair_lag", 3)
            setups.taylor_green(lbm)
            lbm.initialize()
            h = lbm.domains[0].h
            if name == "by hand":
                cfd_amd.lbm.check(L.lbmhip_stream_collide_planes(h, C.c_uint64(0), zero, zero, zero, 4, 36))
                cfd_amd.lbm.check(L.lbmhip_stream_collide_planes(h, C.c_uint64(1), zero, zero, zero, 6, 34))
            else:
                cfd_amd.lbm.check(L.lbmhip_stream_collide_pair_planes(h, C.c_uint64(0), zero, zero, zero, 4, 36))
                assert fused_flag(lbm) == fuse
            got[name] = bits(lbm.domains[0].download_fi()).copy()
    assert np.array_equal(got["one launch"], got["by hand"]) and np.array_equal(got["two launches"], got["by hand"])
    with cfd_amd.LBM(N=(64, 32, 48), nu=0.02, D=(1, 1, 2), devices=[0, 0]) as lbm:
        setups.taylor_green(lbm)
        lbm.initialize()
        h = lbm.domains[0].h
        assert L.lbmhip_stream_collide_pair_planes(h, C.c_uint64(0), zero, zero, zero, 0, 20) != 0
        assert b"halo plane" in L.lbmhip_last_error()
        assert L.lbmhip_stream_collide_pair_planes(h, C.c_uint64(0), zero, zero, zero, 30, 20) != 0
        cfd_amd.lbm.check(L.lbmhip_stream_collide_pair_planes(h, C.c_uint64(0), zero, zero, zero, 2, 24))  # the inner planes of a slab of 26


def test_slabs_of_one_process_pair_their_steps(gpu):
    """cfd_amd.LBM with z-slabs (two and three domains on the one GPU): run() takes the steps two at a time, the inner planes of both in
    one launch per domain, and gives the oracle's multi-domain bits."""
    for D, storage in (((1, 1, 2), "FP32"), ((1, 1, 3), "FP16S")):
        kw = dict(N=(128, 32, 40 * D[2]), nu=0.02, D=D, storage=storage)
        a = cfd_amd.LBM(devices=[0] * D[2], **kw); b = H.OracleLBM(**kw)
        try:
            a.set_tuning("fuse", 1); a.set_tuning("pair_lag", 3)
            setups.taylor_green(a); setups.taylor_green(b)
            a.run(9); b.run(9)
            assert a._pairs_slab_steps() and fused_flag(a) == 1
            for da, db in zip(a.domains, b.domains):
                assert np.array_equal(bits(da.download_fi()), bits(db.fi)), (D, storage)
            a.read_from_device(); b.read_from_device()
            assert np.array_equal(bits(a.rho), bits(b.rho)) and np.array_equal(bits(a.u), bits(b.u))
        finally:
            a.close()


@pytest.mark.parametrize("storage", ["FP32", "FP16S"])
def test_headline_lattice_paired_equals_single_launches(gpu, storage):
    """512^3, the lattice of the headline figure, where pairing is automatic (band by band on each XCD): 200 steps with two steps per
    launch (100 launches of a million workgroups each) leave the bits of 200 single launches in every DDF stream."""
    import zlib
    crcs = {}
    for fuse in (-1, 0):
        with cfd_amd.LBM(N=(512, 512, 512), nu=0.01, storage=storage) as lbm:
            lbm.set_tuning("fuse", fuse); lbm.set_tuning("autotune", 0)
            setups.taylor_green(lbm)
            lbm.run(200)
            assert fused_flag(lbm) == (2 if fuse else 0)
            fi = lbm.domains[0].download_fi()
            n = 512 ** 3
            crcs[fuse] = [zlib.crc32(fi[i * n:(i + 1) * n]) for i in range(19)]
    assert crcs[-1] == crcs[0]


def test_random_lattices_feature_sets_and_geometries(gpu):
    """tools/band_fuzz.py: 60 random combinations of lattice, feature set, form, band height, lag, steps per launch, workgroup size
    and slabs, every one bit-identical to a launch per step."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "band_fuzz.py"), "60", "5"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "60 cases, 0 bad" in r.stdout

