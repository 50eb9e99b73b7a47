"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: flags bit-exact; FP32 runs bit-exact in fi, rho and u (both sides evaluate the same FP32 operation order
with fma only where the reference writes fma); FP16S/FP16C runs compare the raw 16-bit DDFs bit-exactly as well.
"""
import ctypes as C

import numpy as np
import pytest

import cfd_amd
from oracle import harness as H
from tests.cases import CASES

pytestmark = pytest.mark.gpu


def run_pair(case, tuning=None):
    kw = dict(case["kw"])
    a = cfd_amd.LBM(**kw)
    b = H.OracleLBM(**kw)
    if tuning:
        for k, v in tuning.items():
            a.set_tuning(k, v)
    case["setup"](a)
    case["setup"](b)
    return a, b


def compare(a, b, what=""):
    a.read_from_device()
    b.read_from_device()
    assert np.array_equal(a.flags, b.flags), what + " flags differ"
    for da, db in zip(a.domains, b.domains):
        fa, fb = da.download_fi(), db.fi
        halo_free = np.ones(len(fb), dtype=bool)
        assert fa.shape == fb.shape
        same = fa.view(np.uint32 if fa.dtype == np.float32 else np.uint16) == fb.view(np.uint32 if fb.dtype == np.float32 else np.uint16)
        assert same[halo_free].all(), "%s DDFs differ in %d of %d slots" % (what, (~same).sum(), same.size)
    assert np.array_equal(a.rho.view(np.uint32), b.rho.view(np.uint32)), what + " rho differs"
    assert np.array_equal(a.u.view(np.uint32), b.u.view(np.uint32)), what + " u differs"
    if a.F is not None:
        a.update_force_field(); b.update_force_field()
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(a.F.view(np.uint32), b.F.view(np.uint32)), what + " F differs"


@pytest.mark.parametrize("name", sorted(CASES))
def test_parity_against_oracle(gpu, name):
    case = CASES[name]
    a, b = run_pair(case)
    try:
        a.run(1); b.run(1)
        compare(a, b, name + " t=1")
        a.run(case["steps"] - 1); b.run(case["steps"] - 1)
        compare(a, b, name + " t=%d" % case["steps"])
        assert np.isfinite(a.u).all()
    finally:
        a.close()


@pytest.mark.parametrize("cells_per_thread,cells_together", [(1, 1), (2, 1), (4, 1), (2, 2), (4, 2)])
@pytest.mark.parametrize("name", ["tgv32_fp32_srt", "cavity32_fp32_srt", "gui_fp16s_srt", "q27_fp16c_force", "all_fp32_srt_8dom", "all_fp16s_trt_2dom", "q19_fp16c_srt"])
def test_vector_widths_agree(gpu, name, cells_per_thread, cells_together):
    """Every kernel geometry computes the same bits: 1, 2 or 4 cells per lane, collided one at a time or two at a time
    in packed arithmetic."""
    case = CASES[name]
    a, b = run_pair(case, tuning={"cells_per_thread": cells_per_thread, "cells_together": cells_together})
    try:
        a.run(case["steps"]); b.run(case["steps"])
        compare(a, b, "%s V=%d P=%d" % (name, cells_per_thread, cells_together))
    finally:
        a.close()


def test_overlap_matches_plain_exchange(gpu):
    case = CASES["q27_fp32_trt_4slab"]
    a = cfd_amd.LBM(overlap=True, **case["kw"]); b = cfd_amd.LBM(overlap=False, **case["kw"])
    try:
        assert a.overlap and not b.overlap
        case["setup"](a); case["setup"](b)
        a.run(25); b.run(25)
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(a.u, b.u) and np.array_equal(a.rho, b.rho)
        for da, db in zip(a.domains, b.domains):
            assert np.array_equal(da.download_fi(), db.download_fi())
    finally:
        a.close(); b.close()


def test_object_sums_match_host_definition(gpu):
    """object_center_of_mass / object_force / object_torque: sums over the cells flagged TYPE_S|TYPE_X.  The reference
    adds group sums with float atomics in arbitrary order, so the comparison is against the plain definition in double,
    for one domain and for several (where halo cells must not be counted twice)."""
    case = CASES["q27_fp16c_force"]
    for D in ((1, 1, 1), (2, 1, 2)):
        kw = dict(case["kw"], D=D)
        with cfd_amd.LBM(**kw) as lbm:
            case["setup"](lbm)
            lbm.run(30)
            marker = 1 | 64
            force = np.array(lbm.object_force(marker))
            torque = np.array(lbm.object_torque((1.5, -2.0, 0.5), marker))
            com = np.array(lbm.object_center_of_mass(marker))
            lbm.read_from_device()
            sel = lbm.flags == marker
            assert sel.sum() > 500
            Nx, Ny, Nz = lbm.N
            z, y, x = np.meshgrid(np.arange(Nz), np.arange(Ny), np.arange(Nx), indexing="ij")
            pos = np.stack([x + 0.5 - 0.5 * Nx, y + 0.5 - 0.5 * Ny, z + 0.5 - 0.5 * Nz]).astype(np.float64)
            F = lbm.F.astype(np.float64)
            want_force = F[:, sel].sum(axis=1)
            r = pos[:, sel] - np.array([1.5, -2.0, 0.5])[:, None]
            want_torque = np.cross(r.T, F[:, sel].T).sum(axis=0)
            want_com = pos[:, sel].mean(axis=1)
            assert np.abs(want_force).max() > 1e-4
            assert np.allclose(force, want_force, rtol=1e-4, atol=1e-7)
            assert np.allclose(torque, want_torque, rtol=1e-4, atol=1e-6)
            assert np.allclose(com, want_com, rtol=1e-5, atol=1e-4)


RAGGED = {
    # rows that are not a multiple of the vector width, rows shorter than a wave, x/y halos under the wide kernel
    "fp16s_30x14x10": dict(kw=dict(N=(30, 14, 10), nu=0.02, storage="FP16S"), setup="tgv"),
    "fp16c_31x9x7": dict(kw=dict(N=(31, 9, 7), nu=0.02, storage="FP16C", collision="TRT"), setup="tgv"),
    "fp16s_36x10x6_wide": dict(kw=dict(N=(36, 10, 6), nu=0.02, storage="FP16S", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)), setup="channel"),
    "fp16s_xy_halo_all": dict(kw=dict(N=(72, 24, 8), nu=0.01, D=(2, 2, 1), storage="FP16S", collision="TRT",
                                      features=("UPDATE_FIELDS", "VOLUME_FORCE", "MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "FORCE_FIELD", "SUBGRID"), f=(1e-4, 0.0, 0.0)), setup="all"),
    "fp16c_q27_x_halo": dict(kw=dict(N=(80, 12, 12), nu=0.02, D=(2, 1, 1), Q=27, storage="FP16C", features=("EQUILIBRIUM_BOUNDARIES",)), setup="inflow"),
    "fp32_q27_x_halo_v2": dict(kw=dict(N=(44, 12, 12), nu=0.02, D=(2, 1, 1), Q=27, features=("MOVING_BOUNDARIES",)), setup="cavity"),
    "fp32_260x6x6": dict(kw=dict(N=(260, 6, 6), nu=0.02), setup="tgv"),
    "fp16s_1028x4x4_wide": dict(kw=dict(N=(1028, 4, 4), nu=0.02, storage="FP16S"), setup="tgv"),
    # rows of several workgroups per stream, the lattice widths of the 2048^3 configuration
    "fp16c_q27_2048x3x4_force": dict(kw=dict(N=(2048, 3, 4), nu=0.02, Q=27, storage="FP16C", features=("VOLUME_FORCE", "FORCE_FIELD"), f=(1e-5, 0.0, 0.0)), setup="channel"),
    "fp32_1030x5x3_x_halo": dict(kw=dict(N=(2056, 5, 3), nu=0.02, D=(2, 1, 1), collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)), setup="channel"),
    "q15_fp16s_3dom": dict(kw=dict(N=(24, 24, 12), nu=0.02, Q=15, D=(1, 3, 1), storage="FP16S"), setup="tgv"),
    # the feature sets of the reference's setup collection, which have kernel instances of their own
    "aero_fp16s_eq_sgs": dict(kw=dict(N=(40, 24, 20), nu=0.005, storage="FP16S", features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="inflow"),
    "aero_fp16c_eq_sgs": dict(kw=dict(N=(36, 20, 24), nu=0.005, storage="FP16C", features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="inflow"),
    "aero_fp32_eq_sgs_2dom": dict(kw=dict(N=(32, 24, 32), nu=0.005, D=(1, 1, 2), features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="inflow"),
    "aero_fp16s_mb_eq_sgs": dict(kw=dict(N=(48, 24, 24), nu=0.005, storage="FP16S", features=("MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="all"),
    "aero_fp16c_mb_eq_sgs": dict(kw=dict(N=(48, 24, 24), nu=0.005, storage="FP16C", features=("MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="all"),
    "aero_fp32_mb_eq_sgs": dict(kw=dict(N=(48, 24, 24), nu=0.005, features=("MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="all"),
    "aero_fp16s_mb_sgs": dict(kw=dict(N=(32, 32, 32), nu=0.005, storage="FP16S", features=("MOVING_BOUNDARIES", "SUBGRID")), setup="cavity"),
    "aero_fp16c_ff_eq_sgs": dict(kw=dict(N=(40, 24, 24), nu=0.005, storage="FP16C", features=("FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup="sphere"),
    # the small velocity sets with extensions (run-time-feature kernels only)
    "q15_fp32_trt_vf_sgs": dict(kw=dict(N=(28, 20, 16), nu=0.01, Q=15, collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0)), setup="channel"),
    "q15_fp16c_all_2dom": dict(kw=dict(N=(40, 24, 16), nu=0.01, Q=15, D=(2, 1, 1), storage="FP16C",
                                       features=("UPDATE_FIELDS", "VOLUME_FORCE", "MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "FORCE_FIELD", "SUBGRID"), f=(1e-4, 0.0, 0.0)), setup="all"),
    "q9_fp16s_mb_trt": dict(kw=dict(N=(48, 40, 1), nu=0.02, Q=9, storage="FP16S", collision="TRT", features=("MOVING_BOUNDARIES",)), setup="cavity2d"),
    # degenerate lattices: every neighbour of a cell is the cell itself, or the lattice is a line / a sheet
    "fp32_1x1x1": dict(kw=dict(N=(1, 1, 1), nu=0.02), setup="tgv"),
    "fp32_2x3x1_trt": dict(kw=dict(N=(2, 3, 1), nu=0.02, collision="TRT"), setup="tgv"),
    "fp16s_64x1x1": dict(kw=dict(N=(64, 1, 1), nu=0.02, storage="FP16S"), setup="tgv"),
    "fp16c_1x1x40_q27": dict(kw=dict(N=(1, 1, 40), nu=0.02, Q=27, storage="FP16C"), setup="tgv"),
    "fp32_4x1x8_2dom": dict(kw=dict(N=(4, 1, 8), nu=0.02, D=(1, 1, 2), features=("VOLUME_FORCE",), f=(0.0, 0.0, 1e-4)), setup="tgv"),
    "q9_fp16c_4dom": dict(kw=dict(N=(40, 24, 1), nu=0.02, Q=9, D=(2, 2, 1), storage="FP16C"), setup="tgv2d"),
}


def _cavity2d(lbm):
    """2-D lid-driven cavity: solid frame, the top row moves in +x."""
    lbm.flags[...] = 0
    lbm.flags[:, 0, :] = 1; lbm.flags[:, -1, :] = 1; lbm.flags[:, :, 0] = 1; lbm.flags[:, :, -1] = 1
    lbm.u[...] = 0.0
    lbm.u[0, :, -1, :] = 0.08


@pytest.mark.parametrize("cells_together", [0, 2])
@pytest.mark.parametrize("name", sorted(RAGGED))
def test_ragged_shapes_and_halos(gpu, name, cells_together):
    from tests import cases as CS
    from cfd_amd import setups
    case = RAGGED[name]
    setup = {"tgv": setups.taylor_green, "channel": CS._channel_setup, "all": CS._all_features_setup, "inflow": setups.inflow_box,
             "cavity": setups.lid_cavity, "tgv2d": CS._tgv2d, "sphere": CS._sphere_setup, "cavity2d": _cavity2d}[case["setup"]]
    a, b = run_pair(dict(kw=case["kw"], setup=setup), tuning={"cells_together": cells_together} if cells_together else None)
    try:
        a.run(1); b.run(1)
        compare(a, b, name + " t=1")
        a.run(11); b.run(11)
        compare(a, b, name + " t=12")
        assert np.isfinite(a.u).all()
    finally:
        a.close()


def test_geometry_search_is_invisible_in_the_results(gpu):
    """The library tries its launch geometries on the first full-lattice launches of a large enough lattice and keeps the
    fastest; a run with the search and a run with fixed defaults must agree bit for bit, and the search must finish."""
    kw = dict(N=(128, 128, 128), nu=0.01, storage="FP16S", collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(1e-4, 0.0, 0.0))
    from tests import cases as CS
    with cfd_amd.LBM(**kw) as a, cfd_amd.LBM(**kw) as b:
        b.set_tuning("autotune", 0)
        CS._channel_setup(a); CS._channel_setup(b)
        assert a.get_tuning("autotune") == 1 and b.get_tuning("autotune") == 0
        a.run(60); b.run(60)
        assert a.get_tuning("tuned") == 1 and b.get_tuning("tuned") == 0
        assert a.get_tuning("cells_per_thread") in (1, 2, 4) and a.get_tuning("cells_together") in (1, 2) and 0 <= a.get_tuning("blocks_per_cu") <= 8
        a.read_from_device(); b.read_from_device()
        assert np.array_equal(a.u.view(np.uint32), b.u.view(np.uint32)) and np.array_equal(a.rho.view(np.uint32), b.rho.view(np.uint32))
        assert np.array_equal(a.domains[0].download_fi(), b.domains[0].download_fi())


def test_reset_restarts_from_the_host_fields(gpu):
    """LBM::reset (FluidX3D-master/src/lbm.cpp:1498-1500): the next run() initializes again from the host copies of
    rho, u and flags, with the step counter back at 0 - same as a fresh lattice holding those fields."""
    from cfd_amd import setups
    kw = dict(N=(24, 20, 16), nu=0.03, storage="FP16S", collision="TRT")
    with cfd_amd.LBM(**kw) as a, cfd_amd.LBM(**kw) as fresh:
        setups.taylor_green(a)
        a.run(7)                      # odd step count: the parity of t must start over
        a.read_from_device()
        a.u[0] += 0.01
        a.flags[3, 4, 5] = 1
        fresh.rho[...] = a.rho; fresh.u[...] = a.u; fresh.flags[...] = a.flags
        a.reset()
        a.run(5); fresh.run(5)
        assert a.t == 5
        a.read_from_device(); fresh.read_from_device()
        assert np.array_equal(a.u.view(np.uint32), fresh.u.view(np.uint32)) and np.array_equal(a.rho.view(np.uint32), fresh.rho.view(np.uint32))
        assert np.array_equal(a.flags, fresh.flags)
        assert np.array_equal(a.domains[0].download_fi(), fresh.domains[0].download_fi())


@pytest.mark.parametrize("pad", [4160, 131072])
@pytest.mark.parametrize("name", ["all_fp16s_trt_2dom", "all_fp32_srt_8dom", "q27_fp16c_force", "cavity32_fp32_srt", "q27_fp32_trt_4slab"])
def test_padded_stream_stride_is_invisible(gpu, monkeypatch, name, pad):
    """Large domains with halo layers keep their DDF streams a padded stride apart (HBM channel spread); uploads,
    downloads, halo kernels and every lattice kernel must address the streams through that stride.  LBMHIP_PAD forces a
    padding on these small lattices; results and the downloaded DDFs (host layout: streams back to back) stay bit-exact."""
    monkeypatch.setenv("LBMHIP_PAD", str(pad))
    case = CASES[name]
    a, b = run_pair(case)
    try:
        stride = C.c_uint64()
        assert cfd_amd.lib().lbmhip_ddf_stride(a.domains[0].h, C.byref(stride)) == 0 and stride.value == a.domains[0].N + pad
        a.run(case["steps"]); b.run(case["steps"])
        compare(a, b, "%s pad=%d" % (name, pad))
    finally:
        a.close()


def test_stream_stride_default_choice(gpu, monkeypatch):
    monkeypatch.delenv("LBMHIP_PAD", raising=False)
    lib = cfd_amd.lib()
    from cfd_amd import lbm as L
    # streams of at least 64 MiB that are not a multiple of 64 MiB -> 2^17 elements; everything else unpadded
    for dims, D, want_pad in (((64, 64, 64), (1, 1, 2), 0), ((512, 512, 66), (1, 1, 2), 1 << 17), ((512, 512, 64), (1, 1, 2), 0), ((512, 512, 66), (1, 1, 1), 1 << 17), ((256, 256, 66), (1, 1, 1), 0)):
        cfg = L.Config(dims[0], dims[1], dims[2], D[0], D[1], D[2], 0, 0, 0, 19, 0, 0, 0, 0.05, 0)
        dom = L.Domain(cfg)
        try:
            stride = C.c_uint64()
            assert lib.lbmhip_ddf_stride(dom.h, C.byref(stride)) == 0
            n = dims[0] * dims[1] * dims[2]
            assert stride.value - n == want_pad, (dims, D, stride.value - n)
        finally:
            dom.close()


def test_runtime_compiled_kernels(gpu, monkeypatch):
    """A feature set without a precompiled instance is compiled at its first launch (hipRTC) for exactly that set, like the
    reference compiles its OpenCL code with the configured extensions; with LBMHIP_JIT=0 the run-time-feature instance
    runs instead.  Both must give the oracle's bits."""
    case = CASES["all_fp16s_trt_2dom"]
    monkeypatch.delenv("LBMHIP_JIT", raising=False)
    a, ref = run_pair(case)
    a.run(case["steps"])                      # the kernel is chosen at the first launch
    jit_a = a.get_tuning("jit")
    monkeypatch.setenv("LBMHIP_JIT", "0")
    b = cfd_amd.LBM(**case["kw"])
    case["setup"](b)
    try:
        b.run(case["steps"]); ref.run(case["steps"])
        assert jit_a == 1, "hipRTC path not taken (LBMHIP_JIT_VERBOSE=1 prints the reason)"
        assert b.get_tuning("jit") == 0
        compare(a, ref, "run-time compiled")
        compare(b, ref, "run-time features")
    finally:
        a.close(); b.close()
