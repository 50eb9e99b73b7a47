"""Diagnostic: time of the mesh voxelizer on a 512^3 lattice for spheres of growing triangle counts (GPU needed)."""
import os
import sys
import time


sys.path.insert(0, ".")
import cfd_amd
from cfd_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with cfd_amd.LBM((n, n, n), 0.01) as lbm:
    lbm.run(1)  # an initialized lattice: the host copies are not refreshed after every call
    for seg, rings in ((24, 12), (96, 48), (256, 128), (360, 180)):
        m = M.place(*M.uv_sphere(segments=seg, rings=rings), (n, n, n), (n / 2 - 0.5,) * 3, None, 0.6 * n)
        for cull in ("1", "0"):
            os.environ["LBMHIP_VOXEL_CULL"] = cull
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                lbm.voxelize_mesh_on_device(m, 1)
                lbm.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            print("%7d triangles  %s  %s ms" % (m.triangle_number, "boxes" if cull == "1" else "every triangle", " ".join("%.2f" % t for t in ts)), flush=True)
    lbm._read_raw(("flags",))
    print("solid cells", int((lbm.flags == 1).sum()))
