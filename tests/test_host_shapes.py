"""CPU: the geometry predicates and mesh transforms of the C++ host (cfd_amd/host/shapes.hpp, utilities.hpp) against their
geometric definitions evaluated with numpy on a 24^3 lattice."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "shapes.hpp"
#include <cstdio>
int main() {
	const uint N = 24u;
	const float3 p(11.5f, 10.0f, 12.25f), n(3.0f, 8.0f, 4.0f), l(9.0f, 5.0f, 13.0f), r3(7.0f, 4.0f, 9.5f);
	const float3 a(3.0f, 4.0f, 5.0f), b(20.0f, 6.0f, 9.0f), c(8.0f, 19.0f, 14.0f);
	for (uint z = 0u; z < N; z++) for (uint y = 0u; y < N; y++) for (uint x = 0u; x < N; x++) {
		const int bits = (sphere(x, y, z, p, 7.3f) << 0) | (ellipsoid(x, y, z, p, r3) << 1) | (cube(x, y, z, p, 9.0f) << 2) | (cuboid(x, y, z, p, l) << 3)
			| (cylinder(x, y, z, p, n, 5.5f) << 4) | (cone(x, y, z, p, n, 2.0f, 6.0f) << 5) | (pipe(x, y, z, p, n, 5.5f) << 6) | (conepipe(x, y, z, p, n, 2.0f, 6.0f) << 7)
			| (triangle(x, y, z, a, b, c) << 8) | (plane(x, y, z, p, n) << 9) | (torus_x(x, y, z, p, 2.5f, 7.0f) << 10) | (torus_y(x, y, z, p, 2.5f, 7.0f) << 11) | (torus_z(x, y, z, p, 2.5f, 7.0f) << 12);
		std::printf("%d\n", bits);
	}
	Mesh m(2u, float3(1.0f, 2.0f, 3.0f));
	m.p0[0] = float3(0, 0, 0); m.p1[0] = float3(2, 0, 0); m.p2[0] = float3(0, 4, 0);
	m.p0[1] = float3(0, 0, 6); m.p1[1] = float3(2, 0, 6); m.p2[1] = float3(0, 4, 6);
	m.find_bounds();
	m.translate(float3(1.0f, 1.0f, 1.0f)); m.scale(2.0f); m.rotate(float3x3(float3(0, 0, 1), radians(90.0f)));
	std::printf("%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n", m.pmin.x, m.pmin.y, m.pmin.z, m.pmax.x, m.pmax.y, m.pmax.z, m.get_min_size(), m.get_max_size());
	return 0;
}
'''


def test_shapes_and_mesh_transforms(tmp_path):
    src = tmp_path / "shapes_harness.cpp"
    src.write_text(HARNESS)
    exe = str(tmp_path / "shapes_harness")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "cfd_amd", "host"), str(src), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    N = 24
    bits = np.array([int(v) for v in out[:N ** 3]]).reshape(N, N, N)
    z, y, x = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    p = np.array([11.5, 10.0, 12.25]); n = np.array([3.0, 8.0, 4.0]); l = np.array([9.0, 5.0, 13.0]); r3 = np.array([7.0, 4.0, 9.5])
    t = np.stack([x - p[0], y - p[1], z - p[2]]).astype(np.float64)
    r2 = (t ** 2).sum(axis=0)
    L = np.linalg.norm(n); along = np.tensordot(n / L, t, axes=1); radial = np.sqrt(np.maximum(r2 - along ** 2, 0.0))
    within = np.abs(along) <= 0.5 * L
    rc = 2.0 + (6.0 - 2.0) * (along + 0.5 * L) / L
    d = np.sqrt(3.0) / 2.0

    def agree(bit, want, margin):
        """want: boolean field; margin: |distance to the surface| field - points closer than 1e-3 may go either way (float rounding)"""
        got = (bits >> bit) & 1
        clear = margin > 1e-3
        assert np.array_equal(got[clear], want[clear].astype(int)), bit
        assert want.sum() > 20, bit

    agree(0, r2 <= 7.3 ** 2, np.abs(np.sqrt(r2) - 7.3))
    e = ((t / r3[:, None, None, None]) ** 2).sum(axis=0)
    agree(1, e <= 1.0, np.abs(e - 1.0) * 4.0)
    box = lambda h: np.all(np.abs(t) <= h[:, None, None, None], axis=0)
    boxm = lambda h: np.min(np.abs(np.abs(t) - h[:, None, None, None]), axis=0)
    agree(2, box(np.full(3, 4.5)), boxm(np.full(3, 4.5)))
    agree(3, box(0.5 * l), boxm(0.5 * l))
    edge = np.minimum(np.abs(np.abs(along) - 0.5 * L), 1e9)
    agree(4, within & (radial <= 5.5), np.minimum(np.abs(radial - 5.5), edge))
    agree(5, within & (radial <= rc), np.minimum(np.abs(radial - rc), edge))
    agree(6, within & (radial >= 5.5 - d) & (radial <= 5.5 + d), np.minimum(np.minimum(np.abs(radial - 5.5 + d), np.abs(radial - 5.5 - d)), edge))
    agree(7, within & (radial >= rc - d) & (radial <= rc + d), np.minimum(np.minimum(np.abs(radial - rc + d), np.abs(radial - rc - d)), edge))
    agree(9, np.tensordot(n, t, axes=1) <= 0.0, np.abs(np.tensordot(n, t, axes=1)) / L)
    for bit, (i, j) in ((10, (1, 2)), (11, (0, 2)), (12, (0, 1))):   # torus about x, y, z: distance from the ring of radius R
        ring = np.sqrt((np.sqrt(t[i] ** 2 + t[j] ** 2) - 7.0) ** 2 + t[3 - i - j] ** 2)
        agree(bit, (ring >= 2.5 - d) & (ring <= 2.5 + d), np.minimum(np.abs(ring - 2.5 + d), np.abs(ring - 2.5 - d)))
    tri = (bits >> 8) & 1   # a one-cell-thick sheet: every marked point is within sqrt(3)/2 of the triangle's plane, and the sheet is connected enough to count
    a, b, c = np.array([3.0, 4.0, 5.0]), np.array([20.0, 6.0, 9.0]), np.array([8.0, 19.0, 14.0])
    nrm = np.cross(b - a, c - a); nrm /= np.linalg.norm(nrm)
    dist = np.abs((x - a[0]) * nrm[0] + (y - a[1]) * nrm[1] + (z - a[2]) * nrm[2])
    assert tri.sum() > 100 and dist[tri == 1].max() <= d + 1e-4
    # mesh: translate (1,1,1), scale 2 about the centre (2,3,4), rotate 90 degrees about z through the centre
    vals = [float(v) for v in out[N ** 3].split()]
    pts = np.array([[0, 0, 0], [2, 0, 0], [0, 4, 0], [0, 0, 6], [2, 0, 6], [0, 4, 6]], dtype=np.float64) + 1.0
    ctr = np.array([2.0, 3.0, 4.0])
    pts = 2.0 * (pts - ctr) + ctr
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    pts = (pts - ctr) @ rot.T + ctr
    want = list(pts.min(axis=0)) + list(pts.max(axis=0)) + [np.ptp(pts, axis=0).min(), np.ptp(pts, axis=0).max()]
    assert np.allclose(vals, want, atol=1e-3)
