"""GPU: the HIP path against the committed fixtures made from the reference's own kernels, plus size-independent
properties at larger lattices.  All comparisons are bit-exact (CRC of raw bytes)."""
import numpy as np
import pytest

import cfd_amd
from cfd_amd import setups
from tests.cases import CASES, LONG_CASES
from tests.golden.make_golden import crc, summarize
from tests.test_oracle_golden import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_reproduces_reference_fixture(gpu, name):
    case = CASES[name]
    meta, pins, z = load_golden(name)
    with cfd_amd.LBM(**case["kw"]) as lbm:
        case["setup"](lbm)
        lbm.run(case["steps"])
        lbm.read_from_device()
        if lbm.F is not None:
            lbm.update_force_field()
            lbm.read_from_device()
        got = summarize(lbm)
        assert np.array_equal(lbm.flags, z["flags"])
        for k, v in got.items():
            if k.startswith("crc"):
                assert v == pins[k], "%s: %s differs from the reference fixture" % (name, k)
        assert [crc(d.download_fi()) for d in lbm.domains] == pins["crc_fi"]
        if "object" in pins:
            # object_* reductions.  F is bit-identical (crc_F above); what is left is the ORDER of the float additions, which the
            # reference leaves to its atomics.  Tolerance: 1e-5 of the sum of magnitudes (cancellation-safe float32 bound).
            import ctypes as C
            o = pins["object"]
            sel = lbm.flags == o["marker"]
            scale_f = float(np.abs(lbm.F[:, sel]).sum())
            for d, dom in enumerate(lbm.domains):   # per domain, the kernels' own definition (halo cells included, local positions)
                s = (C.c_double * 4)()
                cfd_amd.lbm.check(lbm._L.lbmhip_object_sum(dom.h, 1, C.c_uint8(o["marker"]), None, None, s))
                assert np.allclose(s[:3], o["domain_force"][d], rtol=0, atol=1e-5 * scale_f + 1e-12), (d, s[:3], o["domain_force"][d])
                cfd_amd.lbm.check(lbm._L.lbmhip_object_sum(dom.h, 2, C.c_uint8(o["marker"]), (C.c_float * 3)(*o["center"]), None, s))
                assert np.allclose(s[:3], o["domain_torque"][d], rtol=0, atol=1e-5 * scale_f * max(lbm.N) + 1e-12)
                cfd_amd.lbm.check(lbm._L.lbmhip_object_sum(dom.h, 0, C.c_uint8(o["marker"]), None, None, s))
                assert int(s[3]) == o["domain_cells"][d]
                assert np.allclose(s[:3], o["domain_com"][d], rtol=1e-5, atol=1e-3)
            if len(lbm.domains) == 1:   # one domain: the host totals of the reference are the same quantity
                assert np.allclose(lbm.object_force(o["marker"]), o["force"], rtol=0, atol=1e-5 * scale_f)
                assert np.allclose(lbm.object_torque(tuple(o["center"]), o["marker"]), o["torque"], rtol=0, atol=1e-5 * scale_f * max(lbm.N))
                assert np.allclose(lbm.object_center_of_mass(o["marker"]), o["com"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", sorted(LONG_CASES))
def test_hip_matches_reference_after_1000_steps(gpu, name):
    """The benchmark configurations over the 1000 steps the tolerance is stated for (relative L2 of rho and u at most
    1e-5 for FP32 and 1e-3 for 16-bit DDFs; flags bit-exact), against fixtures from the reference's kernels.  The HIP
    path is in fact bit-identical, which the CRC pins check as well."""
    case = LONG_CASES[name]
    meta, pins, z = load_golden(name)
    assert meta["steps"] == case["steps"] == 1000
    with cfd_amd.LBM(**case["kw"]) as lbm:
        case["setup"](lbm)
        lbm.run(case["steps"])
        lbm.read_from_device()
        assert np.array_equal(lbm.flags, z["flags"])
        for got, want in ((lbm.rho[::4, ::4, ::4], z["rho_sub"]), (lbm.u[:, ::4, ::4, ::4], z["u_sub"])):
            err = np.linalg.norm((got.astype(np.float64) - want).ravel()) / np.linalg.norm(want.astype(np.float64).ravel())
            assert err <= case["tolerance"], (name, err)
        got = summarize(lbm)
        assert got["crc_rho"] == pins["crc_rho"] and got["crc_u"] == pins["crc_u"]
        assert [crc(d.download_fi()) for d in lbm.domains] == pins["crc_fi"]
        assert np.isfinite(lbm.u).all() and float(np.abs(lbm.u).max()) > 1e-4


def _run(N, steps, shift=(0, 0, 0), **kw):
    with cfd_amd.LBM(N, 0.01, **kw) as lbm:
        setups.taylor_green(lbm)
        for name in ("rho",):
            setattr(lbm, name, np.roll(getattr(lbm, name), shift[::-1], axis=(0, 1, 2)))
        lbm.u = np.roll(lbm.u, shift[::-1], axis=(1, 2, 3))
        lbm.run(steps)
        lbm.read_from_device()
        return lbm.rho.copy(), lbm.u.copy()


@pytest.mark.parametrize("N,storage", [((256, 256, 256), "FP32"), ((512, 512, 512), "FP32"), ((256, 256, 256), "FP16S")])
def test_translation_invariance_at_scale(gpu, N, storage):
    """On a periodic lattice a shifted initial state must give the shifted result bit for bit: every cell runs the same
    arithmetic wherever it sits, so any indexing or wrap error at large sizes shows up."""
    steps = 6
    rho0, u0 = _run(N, steps, storage=storage)
    shift = (5, 3, 2)
    rho1, u1 = _run(N, steps, shift=shift, storage=storage)
    assert np.array_equal(np.roll(rho0, shift[::-1], axis=(0, 1, 2)).view(np.uint32), rho1.view(np.uint32))
    assert np.array_equal(np.roll(u0, shift[::-1], axis=(1, 2, 3)).view(np.uint32), u1.view(np.uint32))
    mass = float(rho0.sum(dtype=np.float64)) / rho0.size
    assert abs(mass - 1.0) < 1e-5


def test_decomposition_invariance_at_scale(gpu):
    """1 domain and 2x2x2 domains (all on this GPU) give identical fields: halo pack/exchange/unpack is exact."""
    N = (192, 160, 128)
    rho0, u0 = _run(N, 12)
    rho1, u1 = _run(N, 12, D=(2, 2, 2))
    rho2, u2 = _run(N, 12, D=(1, 1, 4))
    assert np.array_equal(rho0.view(np.uint32), rho1.view(np.uint32)) and np.array_equal(u0.view(np.uint32), u1.view(np.uint32))
    assert np.array_equal(rho0.view(np.uint32), rho2.view(np.uint32)) and np.array_equal(u0.view(np.uint32), u2.view(np.uint32))


@pytest.mark.parametrize("name,kw,slabs", [
    ("channel_512_rows", dict(N=(512, 512, 64), collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0)), 2),
    ("channel_512_rows_fp16s", dict(N=(512, 512, 64), storage="FP16S", collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0)), 4),
    ("q27_force_2048_rows", dict(N=(2048, 512, 16), Q=27, storage="FP16C", features=("VOLUME_FORCE", "FORCE_FIELD"), f=(1e-5, 0.0, 0.0)), 2),
])
def test_benchmark_configurations_do_not_depend_on_the_decomposition(gpu, name, kw, slabs):
    """The channel and the D3Q27 force-field configuration of bench.py at their real row widths (512 and 2048 cells, several
    workgroups per row, padded stream strides on the slabs): one domain and z-slabs must agree bit for bit, boundary force included."""
    def run(D):
        with cfd_amd.LBM(nu=0.01, D=D, **kw) as lbm:
            setups.channel(lbm)
            lbm.run(15)
            if lbm.has_F:
                lbm.update_force_field()
            lbm.read_from_device()
            assert np.isfinite(lbm.u).all() and float(np.abs(lbm.u).max()) > 1e-5
            return lbm.rho.copy(), lbm.u.copy(), (lbm.F.copy() if lbm.has_F else None)
    one, many = run((1, 1, 1)), run((1, 1, slabs))
    assert np.array_equal(one[0].view(np.uint32), many[0].view(np.uint32))
    assert np.array_equal(one[1].view(np.uint32), many[1].view(np.uint32))
    if one[2] is not None:
        assert np.array_equal(one[2].view(np.uint32), many[2].view(np.uint32)) and np.abs(one[2]).max() > 0


# ---- mesh voxelizer on the GPU against the reference fixtures
import os  # noqa: E402

from tests.voxel_cases import VOXEL_CASES, build_mesh  # noqa: E402
from tests.test_oracle_golden import GOLDEN  # noqa: E402


@pytest.mark.parametrize("pad", [0, 4160])
@pytest.mark.parametrize("name", sorted(VOXEL_CASES))
def test_hip_voxelizer_reproduces_reference_fixture(gpu, monkeypatch, name, pad):
    if pad:
        monkeypatch.setenv("LBMHIP_PAD", str(pad))  # DDF streams a padded stride apart: the moving-mesh path writes DDFs
    case = VOXEL_CASES[name]
    z = np.load(os.path.join(GOLDEN, "voxel_" + name + ".npz"), allow_pickle=False)
    with cfd_amd.LBM(case["N"], 0.01, D=case["D"]) as lbm:
        vel = case["vel"] or ((0, 0, 0), (0, 0, 0), (0, 0, 0))
        lbm.write_to_device(("flags", "u"))
        lbm.voxelize_mesh_on_device(build_mesh(case), 1, *vel)
        assert np.array_equal(lbm.flags, z["flags"])
        assert crc(lbm.u) == int(z["crc_u"])


def test_voxelize_stl_then_run(gpu, tmp_path):
    """The GUI's path (main.py:145-198): voxelize an STL, TYPE_E faces, inflow velocity, run."""
    from cfd_amd import mesh as M
    path = str(tmp_path / "body.stl")
    M.write_stl(path, *M.uv_sphere())
    with cfd_amd.LBM((48, 48, 48), 0.01, storage="FP16S", features=("EQUILIBRIUM_BOUNDARIES", "SUBGRID", "UPDATE_FIELDS")) as lbm:
        lbm.voxelize_stl(path, size=0.4 * 48)
        solid = lbm.flags == 1
        assert 3000 < solid.sum() < 4500  # a sphere of diameter 19.2 cells
        lbm.u[0][~solid] = 0.075
        face = np.zeros_like(solid); face[0] = face[-1] = True; face[:, 0] = face[:, -1] = True; face[:, :, 0] = face[:, :, -1] = True
        lbm.flags[face] = 2
        lbm.run(50)
        lbm.read_from_device()
        assert np.isfinite(lbm.u).all() and (lbm.flags[solid] == 1).all()
        assert np.abs(lbm.u[0][~solid & ~face] - 0.075).max() > 1e-3  # the body disturbs the flow


@pytest.mark.parametrize("direction", [0, 1, 2])
@pytest.mark.parametrize("storage,moving", [("FP32", False), ("FP16S", True)])
def test_voxelizer_triangle_boxes_change_nothing(gpu, monkeypatch, direction, storage, moving):
    """The voxelizer skips, per wave, triangles whose projection lies clear of the wave's rays.  Against the plain loop over
    every triangle (the reference's, LBMHIP_VOXEL_CULL=0) on triangle soups that are not even closed surfaces - tiny, huge,
    degenerate, axis-aligned and nearly edge-on triangles - flags, u and DDFs must be the same bits."""
    import ctypes as C
    lib = cfd_amd.lib()
    rng = np.random.RandomState(7 + direction)
    N = (70, 45, 38)
    T = 700
    c = rng.uniform(2.0, 1.0, size=(T, 1, 3)) * 0 + rng.uniform([2, 2, 2], [N[0] - 2, N[1] - 2, N[2] - 2], size=(T, 3))[:, None, :]
    size = np.choose(rng.randint(0, 4, T), [0.02, 0.7, 6.0, 40.0])[:, None, None]
    tri = (c + size * rng.standard_normal((T, 3, 3))).astype(np.float32)
    tri[::9, 1] = tri[::9, 0]                                   # degenerate: two corners coincide
    tri[1::9, :, direction] = tri[1::9, :1, direction]          # faces across the rays, at one depth
    other = (direction + 1) % 3
    tri[2::9, :, other] = tri[2::9, :1, other]                  # seen exactly edge-on by the rays
    tri[3::9, :, other] = tri[3::9, :1, other] + (1e-4 * rng.standard_normal((len(tri[3::9]), 3))).astype(np.float32)  # nearly edge-on
    tri[4::9] = np.round(tri[4::9]) + 0.5                       # corners exactly on rays
    p0, p1, p2 = (np.ascontiguousarray(tri[:, k].ravel()) for k in range(3))
    box = np.concatenate([tri.reshape(-1, 3).min(0) - 2, tri.reshape(-1, 3).max(0) + 2]).astype(np.float32)
    vel = (np.array([30.0, 20.0, 15.0], np.float32), np.array([0.01, 0.0, 0.02], np.float32), np.array([0.001, 0.003, 0.002], np.float32)) if moving else (None, None, None)
    fp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)

    def run(cull):
        monkeypatch.setenv("LBMHIP_VOXEL_CULL", cull)
        with cfd_amd.LBM(N, 0.01, D=(2, 1, 1), storage=storage) as lbm:
            setups.taylor_green(lbm)
            lbm.run(2)
            for dom in lbm.domains:
                cfd_amd.lbm.check(lib.lbmhip_voxelize_mesh(dom.h, C.c_uint32(direction), C.c_uint64(lbm.t + 1), C.c_uint8(1), fp(p0), fp(p1), fp(p2), C.c_uint32(T), fp(box),
                                                           fp(vel[0]), fp(vel[1]), fp(vel[2])))
            lbm._read_raw(("flags", "u"))
            return lbm.flags.copy(), lbm.u.copy(), [dom.download_fi().copy() for dom in lbm.domains]
    with_boxes, plain = run("1"), run("0")
    assert np.array_equal(with_boxes[0], plain[0]) and (plain[0] == 1).sum() > 100
    assert np.array_equal(with_boxes[1].view(np.uint32), plain[1].view(np.uint32))
    for x, y in zip(with_boxes[2], plain[2]):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
