"""CPU: the C restatement (oracle/liblbm_oracle.so) against the golden fixtures made from the reference's kernels.

Every pin is a CRC of raw bytes, so this is a bit-exact check of flags, rho, u, F and the DDF buffers."""
import json
import os

import numpy as np
import pytest

from oracle import harness as H
from tests.cases import CASES
from tests.golden.make_golden import crc, summarize

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return json.loads(str(z["meta"])), json.loads(str(z["pins"])), z


def test_every_case_has_a_fixture():
    for name in CASES:
        assert os.path.exists(os.path.join(GOLDEN, name + ".npz")), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_port_reproduces_reference_fixture(name):
    case = CASES[name]
    meta, pins, z = load_golden(name)
    assert meta["steps"] == case["steps"]
    lbm = H.OracleLBM(**case["kw"])
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
    assert [crc(d.fi) for d in lbm.domains] == pins["crc_fi"]
    assert np.array_equal(lbm.rho[::2, ::2, ::2], z["rho_sub"]) and np.array_equal(lbm.u[:, ::2, ::2, ::2], z["u_sub"])
    assert abs(got["sum_rho"] - pins["sum_rho"]) == 0.0
    if "object" in pins:  # the object_* reduction kernels: same summation order as oracle/_ref, so the floats are identical
        o = pins["object"]
        f32 = lambda v: np.asarray(v, dtype=np.float32)
        assert np.array_equal(lbm.object_center_of_mass(o["marker"]), f32(o["com"]))
        assert np.array_equal(lbm.object_force(o["marker"]), f32(o["force"]))
        assert np.array_equal(lbm.object_torque(tuple(o["center"]), o["marker"]), f32(o["torque"]))
        for d, dom in enumerate(lbm.domains):
            assert np.array_equal(dom.engine.object_sum(1, dom.F, dom.flags, o["marker"])[:3], f32(o["domain_force"][d]))
        assert sum(o["domain_cells"]) > 100 and np.abs(f32(o["force"])).max() > 0


@pytest.mark.parametrize("name", ["tgv32_fp32_srt", "all_fp16s_trt_2dom", "q27_fp16c_force", "q9_fp32_srt"])
def test_port_matches_compiled_reference_step_by_step(name):
    """Direct comparison with oracle/_ref when those libraries are present (build container and GPU box)."""
    if not H.RefEngine.available(name):
        pytest.skip("oracle/_ref/libref_%s.so not built (needs /root/reference)" % name)
    case = CASES[name]
    a = H.OracleLBM(**case["kw"])
    b = H.OracleLBM(engine_factory=lambda local, D: H.RefEngine(name), **case["kw"])
    case["setup"](a); case["setup"](b)
    for _ in range(5):
        a.run(2); b.run(2)
        for da, db in zip(a.domains, b.domains):
            assert np.array_equal(da.fi, db.fi)
    a.read_from_device(); b.read_from_device()
    assert np.array_equal(a.rho.view(np.uint32), b.rho.view(np.uint32)) and np.array_equal(a.u.view(np.uint32), b.u.view(np.uint32))
    assert np.array_equal(a.flags, b.flags)


def test_two_step_bounce_back():
    """A pulse sent into a wall is parked in the solid cell for one step and comes back reversed two steps later
    (implicit full-way bounce-back of the reference: solid cells are skipped and their slots act as a mailbox)."""
    lbm = H.OracleLBM((8, 8, 12), 1.0 / 6.0)  # tau = 1: post-collision = equilibrium, keeps the test readable
    lbm.flags[0] = 1; lbm.flags[-1] = 1
    lbm.u[2, 6] = 0.05  # plane z=6 moves towards +z
    lbm.run(1); lbm.read_from_device()
    m1 = float(lbm.u[2].sum(dtype=np.float64))
    lbm.run(30); lbm.read_from_device()
    assert np.isfinite(lbm.u).all()
    mass = float(lbm.rho[1:-1].sum(dtype=np.float64))
    assert abs(mass - 8 * 8 * 10) < 0.1           # walls only park mass for one step (mailbox slots), they do not create or destroy it
    assert abs(float(lbm.u[2].sum(dtype=np.float64))) < abs(m1)  # the walls have absorbed z-momentum


# ---- mesh voxelizer
from tests.voxel_cases import VOXEL_CASES, voxelize_with_engine  # noqa: E402


@pytest.mark.parametrize("name", sorted(VOXEL_CASES))
def test_port_voxelizer_reproduces_reference_fixture(name):
    case = VOXEL_CASES[name]
    z = np.load(os.path.join(GOLDEN, "voxel_" + name + ".npz"), allow_pickle=False)
    flags, u = voxelize_with_engine(case, lambda local, D, d: H.PortEngine(local, D, 19, "FP32", "SRT", (), 0.01))
    assert flags.sum() > 100
    assert np.array_equal(flags, z["flags"])
    assert crc(u) == int(z["crc_u"])


def test_stl_round_trip(tmp_path):
    from cfd_amd import mesh as M
    p0, p1, p2 = M.torus()
    path = str(tmp_path / "torus.stl")
    M.write_stl(path, p0, p1, p2)
    a = M.read_stl(path, (40, 40, 40), (19.5, 19.5, 19.5), None, 30.0)
    b = M.place(p0, p1, p2, (40, 40, 40), (19.5, 19.5, 19.5), None, 30.0)
    assert a.triangle_number == 1024 and np.array_equal(a.p0, b.p0) and np.array_equal(a.p2, b.p2)
    assert abs(float(a.get_max_size()) - 30.0) < 1e-4
    assert np.allclose(0.5 * (a.pmin + a.pmax), (19.5, 19.5, 19.5), atol=1e-4)
    with pytest.raises(ValueError):
        open(path, "ab").write(b"x"); M.read_stl(path, (40, 40, 40), (0, 0, 0))
