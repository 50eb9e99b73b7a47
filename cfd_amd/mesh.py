"""Triangle meshes on the host: binary STL reading, placement in the lattice, and the voxelizer's launch parameters.

Follows the reference's Mesh / read_stl (FluidX3D-master/src/utilities.hpp:4440-4581) and the parameter packing of
LBM_Domain::voxelize_mesh_on_device (FluidX3D-master/src/lbm.cpp:344-412).  All arithmetic in float32, in the
reference's order, so that the triangles handed to the voxelizer are the same numbers.
"""
import struct

import numpy as np

f32 = np.float32


class Mesh:
    """p0, p1, p2: float32 arrays [triangles, 3] (the corners of every triangle, lattice coordinates)."""

    def __init__(self, p0, p1, p2, center=(0.0, 0.0, 0.0)):
        self.p0 = np.ascontiguousarray(p0, dtype=f32)
        self.p1 = np.ascontiguousarray(p1, dtype=f32)
        self.p2 = np.ascontiguousarray(p2, dtype=f32)
        self.center = np.asarray(center, dtype=f32)
        self.find_bounds()

    @property
    def triangle_number(self):
        return self.p0.shape[0]

    def find_bounds(self):
        allp = np.stack([self.p0, self.p1, self.p2])
        self.pmin = allp.min(axis=(0, 1)).astype(f32)
        self.pmax = allp.max(axis=(0, 1)).astype(f32)

    def get_max_size(self):
        return f32((self.pmax - self.pmin).max())

    def get_scale_for_box_fit(self, box_size):
        return f32((np.asarray(box_size, dtype=f32) / (self.pmax - self.pmin)).min())


def rotation_matrix(axis, angle):
    """Rotation about `axis` by `angle` radians (Rodrigues), float32."""
    n = np.asarray(axis, dtype=np.float64)
    n = n / np.linalg.norm(n)
    c, s = np.cos(angle), np.sin(angle)
    t = 1.0 - c
    x, y, z = n
    return np.array([[t * x * x + c, t * x * y - s * z, t * x * z + s * y], [t * y * x + s * z, t * y * y + c, t * y * z - s * x],
                     [t * z * x - s * y, t * z * y + s * x, t * z * z + c]], dtype=f32)


def place(p0, p1, p2, box_size, center, rotation=None, size=0.0, reposition=True):
    """read_stl_raw after the file has been parsed (utilities.hpp:4550-4571): rotate, then rescale (size 0: largest
    that fits the box; >0: longest bounding-box side = size; <0: factor -size), re-centre on `center`."""
    R = np.eye(3, dtype=f32) if rotation is None else np.asarray(rotation, dtype=f32)
    rot = lambda p: (np.asarray(p, dtype=f32) @ R.T).astype(f32)
    m = Mesh(rot(p0), rot(p1), rot(p2), center)
    if size == 0.0:
        scale = m.get_scale_for_box_fit(box_size)
    elif size > 0.0:
        scale = f32(size) / m.get_max_size()
    else:
        scale = f32(-size)
    offset = f32(-0.5) * (m.pmin + m.pmax) if reposition else np.zeros(3, dtype=f32)
    c = np.asarray(center, dtype=f32)
    out = Mesh(c + scale * (offset + m.p0), c + scale * (offset + m.p1), c + scale * (offset + m.p2), c)
    return out


def read_stl(path, box_size, center, rotation=None, size=0.0):
    """Binary STL: 80-byte header, uint32 triangle count, 50 bytes per triangle (normal, 3 corners, attribute)."""
    raw = open(path, "rb").read()
    if len(raw) < 84:
        raise ValueError('File "%s" is corrupt!' % path)
    n = struct.unpack_from("<I", raw, 80)[0]
    if n == 0 or len(raw) != 84 + 50 * n:
        raise ValueError('File "%s" is corrupt or unsupported! Only binary .stl files are supported.' % path)
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("p", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return place(rec["p"][:, 0], rec["p"][:, 1], rec["p"][:, 2], box_size, center, rotation, size)


def write_stl(path, p0, p1, p2):
    p0, p1, p2 = (np.asarray(p, dtype=f32) for p in (p0, p1, p2))
    rec = np.zeros(p0.shape[0], dtype=np.dtype([("n", "<f4", 3), ("p", "<f4", (3, 3)), ("a", "<u2")]))
    rec["p"][:, 0], rec["p"][:, 1], rec["p"][:, 2] = p0, p1, p2
    nrm = np.cross(p1 - p0, p2 - p0)
    rec["n"] = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    with open(path, "wb") as f:
        f.write(b"cfd_amd test mesh".ljust(80, b" "))
        f.write(struct.pack("<I", p0.shape[0]))
        f.write(rec.tobytes())


def voxelize_parameters(mesh, rotation_center=(0, 0, 0), linear_velocity=(0, 0, 0), rotational_velocity=(0, 0, 0)):
    """Bounding box widened by 2 cells and the ray direction (lbm.cpp:352-398): without rotation the axis with the
    smallest bounding-box cross-section, otherwise the axis closest to the rotation axis."""
    lo = (mesh.pmin - f32(2.0)).astype(f32)
    hi = (mesh.pmax + f32(2.0)).astype(f32)
    rot = np.asarray(rotational_velocity, dtype=f32)
    if float(np.linalg.norm(rot)) == 0.0:
        e = hi - lo
        v = [e[1] * e[2], e[2] * e[0], e[0] * e[1]]
        direction = 0
        for i in (1, 2):
            if v[i] < v[direction]:
                direction = i
    else:
        v = np.abs(rot)
        direction = 0
        for i in (1, 2):
            if v[i] > v[direction]:
                direction = i
    bbox = np.concatenate([lo, hi]).astype(f32)
    return direction, bbox, np.asarray(rotation_center, dtype=f32), np.asarray(linear_velocity, dtype=f32), rot


def packed_parameters(mesh, rotation_center=(0, 0, 0), linear_velocity=(0, 0, 0), rotational_velocity=(0, 0, 0)):
    """The 16 floats the reference hands to its kernel (lbm.cpp:359-374); used by the test oracles."""
    direction, bbox, rc, lin, rot = voxelize_parameters(mesh, rotation_center, linear_velocity, rotational_velocity)
    params = np.zeros(16, dtype=f32)
    params[0:1] = np.array([mesh.triangle_number], dtype=np.uint32).view(f32)
    params[1:7], params[7:10], params[10:13], params[13:16] = bbox, rc, lin, rot
    return direction, params


# ---- procedural closed meshes (tests and examples)
def uv_sphere(radius=1.0, segments=24, rings=12):
    th = np.linspace(0.0, np.pi, rings + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, segments + 1)
    P = lambda i, j: np.array([np.sin(th[i]) * np.cos(ph[j]), np.sin(th[i]) * np.sin(ph[j]), np.cos(th[i])]) * radius
    t = []
    for i in range(rings):
        for j in range(segments):
            a, b, c, d = P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)
            if i > 0:
                t.append((a, b, d))
            if i < rings - 1:
                t.append((b, c, d))
    t = np.array(t, dtype=f32)
    return t[:, 0], t[:, 1], t[:, 2]


def torus(R=1.0, r=0.35, segments=32, sides=16):
    u = np.linspace(0.0, 2.0 * np.pi, segments + 1)
    v = np.linspace(0.0, 2.0 * np.pi, sides + 1)
    P = lambda i, j: np.array([(R + r * np.cos(v[j])) * np.cos(u[i]), (R + r * np.cos(v[j])) * np.sin(u[i]), r * np.sin(v[j])])
    t = []
    for i in range(segments):
        for j in range(sides):
            a, b, c, d = P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)
            t.append((a, b, c)); t.append((a, c, d))
    t = np.array(t, dtype=f32)
    return t[:, 0], t[:, 1], t[:, 2]


def box(sx=1.0, sy=1.0, sz=1.0):
    x, y, z = sx / 2, sy / 2, sz / 2
    v = np.array([[-x, -y, -z], [x, -y, -z], [x, y, -z], [-x, y, -z], [-x, -y, z], [x, -y, z], [x, y, z], [-x, y, z]], dtype=f32)
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]
    t = []
    for a, b, c, d in quads:
        t.append((v[a], v[b], v[c])); t.append((v[a], v[c], v[d]))
    t = np.array(t, dtype=f32)
    return t[:, 0], t[:, 1], t[:, 2]
