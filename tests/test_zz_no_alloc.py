"""lbmhip_voxelize_mesh_static (triangles staged in the library's fixed device buffer, no allocation) against
lbmhip_voxelize_mesh (buffer owned by the domain): identical flags, u and DDFs, on two calls in a row, the second with a
smaller mesh, and once more after the buffer has held a larger mesh.  Both go through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import cfd_amd
from cfd_amd import mesh as M
from tests.voxel_cases import VOXEL_CASES, build_mesh

pytestmark = pytest.mark.gpu

VEL = ((15.5, 15.5, 15.5), (0.01, 0.0, 0.02), (0.001, 0.003, 0.002))


def _state(lbm):
    lbm._read_raw(("flags", "u"))
    return lbm.flags.copy(), lbm.u.copy(), [dom.download_fi().copy() for dom in lbm.domains]


def _voxelize(mesh_a, mesh_b, N, D, static, storage):
    """Mesh a (moving: writes flags, u and DDFs) with flag 1, then the smaller mesh b (at rest) with flag 65."""
    with cfd_amd.LBM(N, 0.01, D=D, storage=storage) as lbm:
        lbm.write_to_device(("flags", "u"))
        lbm.voxelize_mesh_on_device(mesh_a, 1, *VEL, static=static)
        first = _state(lbm)
        lbm.voxelize_mesh_on_device(mesh_b, 65, static=static)
        second = _state(lbm)
    return first, second


def _same(a, b):
    assert np.array_equal(a[0], b[0]), "flags differ"
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), "u differs"
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), "DDFs differ"


@pytest.mark.parametrize("storage", ["FP32", "FP16C"])
def test_static_buffer_voxelizer_equals_allocating_voxelizer(gpu, storage):
    N, D = (32, 32, 32), (2, 2, 2)
    big = M.place(*M.uv_sphere(segments=48, rings=24), N, (15.5, 15.5, 15.5), None, 18.0)     # 2208 triangles
    small = M.place(*M.box(1.0, 0.6, 0.3), N, (9.5, 20.5, 12.5), M.rotation_matrix((0.2, 1.0, 0.5), 1.0), 9.0)  # 12 triangles
    assert small.triangle_number < big.triangle_number
    want = _voxelize(big, small, N, D, False, storage)
    got = _voxelize(big, small, N, D, True, storage)
    _same(got[0], want[0])
    _same(got[1], want[1])
    assert (want[0][0] == 1).sum() > 1000 and (want[1][0] == 65).sum() > 50 and np.abs(want[0][1]).max() > 0


def test_static_buffer_voxelizer_reproduces_reference_fixture(gpu):
    import os
    from tests.test_oracle_golden import GOLDEN
    from tests.golden.make_golden import crc
    for name in ("sphere_spin_8dom", "box_fit_2dom"):  # 528 triangles, then 12
        case = VOXEL_CASES[name]
        z = np.load(os.path.join(GOLDEN, "voxel_" + name + ".npz"), allow_pickle=False)
        with cfd_amd.LBM(case["N"], 0.01, D=case["D"]) as lbm:
            vel = case["vel"] or ((0, 0, 0), (0, 0, 0), (0, 0, 0))
            lbm.write_to_device(("flags", "u"))
            lbm.voxelize_mesh_on_device(build_mesh(case), 1, *vel, static=True)
            assert np.array_equal(lbm.flags, z["flags"])
            assert crc(lbm.u) == int(z["crc_u"])


def test_static_buffer_voxelizer_rejects_meshes_beyond_its_capacity(gpu):
    lib = cfd_amd.lib()
    limit = 131072  # LBMHIP_STATIC_MESH_TRIANGLES
    with cfd_amd.LBM((16, 16, 16), 0.01) as lbm:
        n = limit + 1
        tri = np.zeros(3 * n, dtype=np.float32)
        box = np.array([0, 0, 0, 8, 8, 8], dtype=np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        h = lbm.domains[0].h
        rc = lib.lbmhip_voxelize_mesh_static(h, C.c_uint32(0), C.c_uint64(1), C.c_uint8(1), p(tri), p(tri), p(tri), C.c_uint32(n), p(box), None, None, None)
        assert rc != 0 and b"exceeds the static buffer" in lib.lbmhip_last_error()
        lbm._read_raw(("flags",))
        assert not lbm.flags.any()
