"""Voxelizer parity cases: procedural closed meshes placed in small lattices (single and split domains)."""
import numpy as np

from cfd_amd import decomposition
from cfd_amd import mesh as M

VOXEL_CASES = {
    "sphere32": dict(N=(32, 32, 32), D=(1, 1, 1), tri="uv_sphere", rotation=None, size=20.0, center=(15.5, 15.5, 15.5), vel=None),
    "torus_rot": dict(N=(40, 36, 28), D=(1, 1, 1), tri="torus", rotation=((1.0, 1.0, 0.0), 0.6), size=26.0, center=(19.5, 17.5, 13.5), vel=None),
    "box_fit_2dom": dict(N=(24, 24, 48), D=(1, 1, 2), tri="box", rotation=((0.2, 1.0, 0.5), 1.0), size=0.0, center=(11.5, 11.5, 23.5), vel=None),
    "sphere_spin_8dom": dict(N=(32, 32, 32), D=(2, 2, 2), tri="uv_sphere", rotation=None, size=18.0, center=(15.5, 15.5, 15.5),
                             vel=((15.5, 15.5, 15.5), (0.01, 0.0, 0.02), (0.001, 0.003, 0.002))),
}


def build_mesh(case):
    tri = {"uv_sphere": M.uv_sphere, "torus": M.torus, "box": lambda: M.box(1.0, 0.6, 0.3)}[case["tri"]]()
    rot = M.rotation_matrix(*case["rotation"]) if case["rotation"] else None
    return M.place(*tri, case["N"], case["center"], rot, case["size"])


def voxelize_with_engine(case, engine_factory):
    """Per-domain voxelization with a CPU engine (oracle side); returns global flags and u like LBM.voxelize_mesh_on_device."""
    N, D = case["N"], case["D"]
    m = build_mesh(case)
    vel = case["vel"] or ((0, 0, 0), (0, 0, 0), (0, 0, 0))
    direction, params = M.packed_parameters(m, *vel)
    local = decomposition.local_size(N, D)
    n = local[0] * local[1] * local[2]
    flags = np.zeros((N[2], N[1], N[0]), dtype=np.uint8)
    u = np.zeros((3, N[2], N[1], N[0]), dtype=np.float32)
    for d in range(D[0] * D[1] * D[2]):
        engine = engine_factory(local, D, d)
        fi = np.zeros(19 * n, dtype=np.float32)
        ul = np.zeros(3 * n, dtype=np.float32)
        fl = np.zeros(n, dtype=np.uint8)
        engine.voxelize_mesh(decomposition.domain_offset(d, N, D), direction, fi, ul, fl, 1, 1, m.p0.ravel().copy(), m.p1.ravel().copy(), m.p2.ravel().copy(), params)
        g, l = decomposition.slices(d, N, D)
        flags[g] = fl.reshape(local[2], local[1], local[0])[l]
        u[(slice(None),) + g] = ul.reshape(3, local[2], local[1], local[0])[(slice(None),) + l]
    return flags, u
