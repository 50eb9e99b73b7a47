"""Test infrastructure: drive the CPU oracles through the reference's host sequencing.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.

Two engines share one interface:
  * PortEngine - oracle/liblbm_oracle.so, the C restatement (runtime-configured);
  * RefEngine  - oracle/_ref/libref_<tag>.so, the reference's own kernels compiled for the host
                 (one library per configuration, constants baked in; built by oracle/ref/build_ref.py).
OracleLBM mirrors the reference's LBM class (FluidX3D-master/src/lbm.cpp:1038-1124, 1354-1459,
2120-2182): global fields, D domains with 1-cell halos, initialize / do_time_step / communicate_*.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(HERE, "liblbm_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

STORAGE = {"FP32": 0, "FP16S": 1, "FP16C": 2}
COLLISION = {"SRT": 0, "TRT": 1}
FEATURE_BITS = {"UPDATE_FIELDS": 1, "VOLUME_FORCE": 2, "MOVING_BOUNDARIES": 4, "EQUILIBRIUM_BOUNDARIES": 8,
                "FORCE_FIELD": 16, "SUBGRID": 32}
TRANSFERS = {9: 3, 15: 5, 19: 5, 27: 9}
TYPE_S, TYPE_E, TYPE_MS, TYPE_X, TYPE_Y = 1, 2, 3, 64, 128


def feature_mask(features):
    m = 0
    for f in features:
        m |= FEATURE_BITS[f]
    return m


def relaxation_rate(nu):
    nu = np.float32(nu)
    tau = np.float32(np.float32(3.0) * nu) + np.float32(0.5)
    return np.float32(np.float32(1.0) / np.float32(tau))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("Nx", "Ny", "Nz", "Dx", "Dy", "Dz", "Q", "storage", "collision", "features")] + [("w", C.c_float)]


class PortEngine:
    """The C restatement, configured for one domain shape."""
    kind = "port"

    def __init__(self, N, D, Q, storage, collision, features, nu):
        if not os.path.exists(PORT_LIB):
            raise FileNotFoundError(PORT_LIB + " (run make -C oracle)")
        self.lib = C.CDLL(PORT_LIB)
        self.cfg = _Cfg(N[0], N[1], N[2], D[0], D[1], D[2], Q, STORAGE[storage], COLLISION[collision], feature_mask(features), float(relaxation_rate(nu)))
        self.c = C.byref(self.cfg)

    def initialize(self, fi, rho, u, flags):
        self.lib.oracle_initialize(self.c, _p(fi), _p(rho), _p(u), _p(flags))

    def stream_collide(self, fi, rho, u, flags, t, fx, fy, fz, F=None, threads=1):
        self.lib.oracle_stream_collide(self.c, _p(fi), _p(rho), _p(u), _p(flags), C.c_uint64(t), C.c_float(fx), C.c_float(fy), C.c_float(fz), _p(F))

    def update_fields(self, fi, rho, u, flags, t, fx, fy, fz, F=None):
        self.lib.oracle_update_fields(self.c, _p(fi), _p(rho), _p(u), _p(flags), C.c_uint64(t), C.c_float(fx), C.c_float(fy), C.c_float(fz), _p(F))

    def update_force_field(self, fi, flags, t, F):
        self.lib.oracle_update_force_field(self.c, _p(fi), _p(flags), C.c_uint64(t), _p(F))

    def update_moving_boundaries(self, u, flags):
        self.lib.oracle_update_moving_boundaries(self.c, _p(u), _p(flags))

    def object_sum(self, what, F, flags, marker, center=(0.0, 0.0, 0.0)):
        out = np.zeros(4, dtype=np.float32)
        self.lib.oracle_object_sum(self.c, int(what), _p(F), _p(flags), C.c_uint8(marker), C.c_float(center[0]), C.c_float(center[1]), C.c_float(center[2]), _p(out))
        return out

    def voxelize_mesh(self, O, direction, fi, u, flags, t, flag, p0, p1, p2, params):
        off = (C.c_int * 3)(*O)
        self.lib.oracle_voxelize_mesh(self.c, off, C.c_uint32(direction), _p(fi), _p(u), _p(flags), C.c_uint64(t), C.c_uint8(flag), _p(p0), _p(p1), _p(p2), _p(params))

    def transfer_fi(self, insert, axis, t, p, m, fi):
        self.lib.oracle_transfer_fi(self.c, int(insert), C.c_uint32(axis), C.c_uint64(t), _p(p), _p(m), _p(fi))

    def transfer_rho_u_flags(self, insert, axis, t, p, m, rho, u, flags):
        self.lib.oracle_transfer_rho_u_flags(self.c, int(insert), C.c_uint32(axis), _p(p), _p(m), _p(rho), _p(u), _p(flags))

    def transfer_F(self, insert, axis, t, p, m, F):
        self.lib.oracle_transfer_F(self.c, int(insert), C.c_uint32(axis), _p(p), _p(m), _p(F))


class RefEngine:
    """The reference's kernels compiled for the host; everything is baked into the library."""
    kind = "reference"

    def __init__(self, tag):
        path = os.path.join(REF_DIR, "libref_%s.so" % tag)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        info = (C.c_int * 10)()
        self.lib.ref_info(info)
        self.N, self.D, self.Q, self.transfers, self.fpxx_bytes, self.features = tuple(info[0:3]), tuple(info[3:6]), info[6], info[7], info[8], info[9]

    @staticmethod
    def available(tag):
        return os.path.exists(os.path.join(REF_DIR, "libref_%s.so" % tag))

    def initialize(self, fi, rho, u, flags, threads=1):
        self.lib.ref_initialize(_p(fi), _p(rho), _p(u), _p(flags), int(threads))

    def stream_collide(self, fi, rho, u, flags, t, fx, fy, fz, F=None, threads=1):
        self.lib.ref_stream_collide(_p(fi), _p(rho), _p(u), _p(flags), C.c_uint64(t), C.c_float(fx), C.c_float(fy), C.c_float(fz), _p(F), int(threads))

    def update_fields(self, fi, rho, u, flags, t, fx, fy, fz, F=None, threads=1):
        self.lib.ref_update_fields(_p(fi), _p(rho), _p(u), _p(flags), C.c_uint64(t), C.c_float(fx), C.c_float(fy), C.c_float(fz), _p(F), int(threads))

    def update_force_field(self, fi, flags, t, F):
        assert self.lib.ref_update_force_field(_p(fi), _p(flags), C.c_uint64(t), _p(F), 1) == 0

    def update_moving_boundaries(self, u, flags):
        assert self.lib.ref_update_moving_boundaries(_p(u), _p(flags), 1) == 0

    def object_sum(self, what, F, flags, marker, center=(0.0, 0.0, 0.0)):
        out = np.zeros(4, dtype=np.float32)
        assert self.lib.ref_object_sum(int(what), _p(F), _p(flags), C.c_uint8(marker), C.c_float(center[0]), C.c_float(center[1]), C.c_float(center[2]), _p(out)) == 0
        return out

    def voxelize_mesh(self, O, direction, fi, u, flags, t, flag, p0, p1, p2, params):
        self.lib.ref_voxelize_mesh(C.c_uint32(direction), _p(fi), _p(u), _p(flags), C.c_uint64(t), C.c_uint8(flag), _p(p0), _p(p1), _p(p2), _p(params))

    def transfer_fi(self, insert, axis, t, p, m, fi):
        self.lib.ref_transfer_fi(int(insert), C.c_uint32(axis), C.c_uint64(t), _p(p), _p(m), _p(fi))

    def transfer_rho_u_flags(self, insert, axis, t, p, m, rho, u, flags):
        self.lib.ref_transfer_rho_u_flags(int(insert), C.c_uint32(axis), C.c_uint64(t), _p(p), _p(m), _p(rho), _p(u), _p(flags))

    def transfer_F(self, insert, axis, t, p, m, F):
        assert self.lib.ref_transfer_F(int(insert), C.c_uint32(axis), C.c_uint64(t), _p(p), _p(m), _p(F)) == 0


class _Domain:
    def __init__(self, engine, N, Q, storage, has_F, has_halo):
        self.engine = engine
        self.N = N
        n = N[0] * N[1] * N[2]
        self.n = n
        self.fi = np.zeros(Q * n, dtype=np.float32 if storage == "FP32" else np.uint16)
        self.rho = np.ones(n, dtype=np.float32)
        self.u = np.zeros(3 * n, dtype=np.float32)
        self.flags = np.zeros(n, dtype=np.uint8)
        self.F = np.zeros(3 * n, dtype=np.float32) if has_F else None
        if has_halo:
            amax = max(N[1] * N[2], N[2] * N[0], N[0] * N[1])
            per_cell = max(TRANSFERS[Q] * (4 if storage == "FP32" else 2), 17)
            self.buf_p = np.zeros(amax * per_cell, dtype=np.uint8)
            self.buf_m = np.zeros(amax * per_cell, dtype=np.uint8)


class OracleLBM:
    """Host sequencing of the reference's LBM class over a CPU engine.

    Global fields are numpy arrays indexed [z, y, x] (u and F: [component, z, y, x]); fill them, then run().
    engine_factory(local_N, D) returns the engine for one domain shape.
    """

    def __init__(self, N, nu, D=(1, 1, 1), Q=19, storage="FP32", collision="SRT", features=(), f=(0.0, 0.0, 0.0), engine_factory=None):
        self.N = tuple((N[k] // D[k]) * D[k] for k in range(3))
        self.D, self.Q, self.storage, self.collision, self.features, self.nu = tuple(D), Q, storage, collision, tuple(features), nu
        self.fx, self.fy, self.fz = f
        self.H = tuple(1 if d > 1 else 0 for d in D)
        self.local = tuple(self.N[k] // D[k] + 2 * self.H[k] for k in range(3))
        self.has_F = "FORCE_FIELD" in features
        if engine_factory is None:
            engine_factory = lambda local, D_: PortEngine(local, D_, Q, storage, collision, features, nu)
        engine = engine_factory(self.local, self.D)
        nd = D[0] * D[1] * D[2]
        self.domains = [_Domain(engine, self.local, Q, storage, self.has_F, nd > 1) for _ in range(nd)]
        Nx, Ny, Nz = self.N
        self.rho = np.ones((Nz, Ny, Nx), dtype=np.float32)
        self.u = np.zeros((3, Nz, Ny, Nx), dtype=np.float32)
        self.flags = np.zeros((Nz, Ny, Nx), dtype=np.uint8)
        self.F = np.zeros((3, Nz, Ny, Nx), dtype=np.float32) if self.has_F else None
        self.t = 0
        self.initialized = False
        self.t_last_update_fields = None

    # ---- global <-> per-domain mapping (FluidX3D-master/src/lbm.hpp:265-288)
    def _slices(self, d):
        Dx, Dy, Dz = self.D
        dx, dy, dz = (d % (Dx * Dy)) % Dx, (d % (Dx * Dy)) // Dx, d // (Dx * Dy)
        n = tuple(self.N[k] // self.D[k] for k in range(3))
        glob = (slice(dz * n[2], (dz + 1) * n[2]), slice(dy * n[1], (dy + 1) * n[1]), slice(dx * n[0], (dx + 1) * n[0]))
        loc = (slice(self.H[2], self.H[2] + n[2]), slice(self.H[1], self.H[1] + n[1]), slice(self.H[0], self.H[0] + n[0]))
        return glob, loc

    def _view(self, dom, name, comps):
        Lx, Ly, Lz = self.local
        a = getattr(dom, name)
        return a.reshape((comps, Lz, Ly, Lx)) if comps > 1 else a.reshape((Lz, Ly, Lx))

    def _scatter(self):
        for d, dom in enumerate(self.domains):
            g, l = self._slices(d)
            self._view(dom, "rho", 1)[l] = self.rho[g]
            self._view(dom, "flags", 1)[l] = self.flags[g]
            self._view(dom, "u", 3)[(slice(None),) + l] = self.u[(slice(None),) + g]
            if self.has_F:
                self._view(dom, "F", 3)[(slice(None),) + l] = self.F[(slice(None),) + g]

    def _gather(self):
        for d, dom in enumerate(self.domains):
            g, l = self._slices(d)
            self.rho[g] = self._view(dom, "rho", 1)[l]
            self.flags[g] = self._view(dom, "flags", 1)[l]
            self.u[(slice(None),) + g] = self._view(dom, "u", 3)[(slice(None),) + l]
            if self.has_F:
                self.F[(slice(None),) + g] = self._view(dom, "F", 3)[(slice(None),) + l]

    # ---- halo exchange (FluidX3D-master/src/lbm.cpp:2120-2182)
    def _communicate(self, field):
        Dx, Dy, Dz = self.D
        for axis in range(3):
            if self.D[axis] <= 1:
                continue
            for dom in self.domains:
                self._transfer(dom, field, 0, axis)
            for d, dom in enumerate(self.domains):
                x, y, z = (d % (Dx * Dy)) % Dx, (d % (Dx * Dy)) // Dx, d // (Dx * Dy)
                c = [x, y, z]
                c[axis] = (c[axis] + 1) % self.D[axis]
                dp = self.domains[c[0] + (c[1] + c[2] * Dy) * Dx]
                dom.buf_p, dp.buf_m = dp.buf_m, dom.buf_p
            for dom in self.domains:
                self._transfer(dom, field, 1, axis)

    def _transfer(self, dom, field, insert, axis):
        e = dom.engine
        if field == "fi":
            e.transfer_fi(insert, axis, self.t, dom.buf_p, dom.buf_m, dom.fi)
        elif field == "rho_u_flags":
            e.transfer_rho_u_flags(insert, axis, self.t, dom.buf_p, dom.buf_m, dom.rho, dom.u, dom.flags)
        elif field == "F":
            e.transfer_F(insert, axis, self.t, dom.buf_p, dom.buf_m, dom.F)

    # ---- FluidX3D-master/src/lbm.cpp:1354-1409
    def initialize(self):
        self._scatter()
        if self.has_F:
            self._communicate("F")
        self.t = 1
        self._communicate("rho_u_flags")
        for dom in self.domains:
            dom.engine.initialize(dom.fi, dom.rho, dom.u, dom.flags)
        self._communicate("rho_u_flags")
        self._communicate("fi")
        self.t = 0
        self.initialized = True
        self.t_last_update_fields = 0 if "UPDATE_FIELDS" in self.features else None

    # ---- FluidX3D-master/src/lbm.cpp:1411-1489
    def run(self, steps, threads=1):
        if not self.initialized:
            self.initialize()
        for _ in range(steps):
            for dom in self.domains:
                dom.engine.stream_collide(dom.fi, dom.rho, dom.u, dom.flags, self.t, self.fx, self.fy, self.fz, dom.F, threads=threads)
            self._communicate("fi")
            self.t += 1
            if "UPDATE_FIELDS" in self.features:
                self.t_last_update_fields = self.t

    # ---- FluidX3D-master/src/lbm.cpp:238-246, 1491-1496; lbm.hpp:390-396
    def update_fields(self):
        if "UPDATE_FIELDS" not in self.features and self.t != self.t_last_update_fields:
            for dom in self.domains:
                dom.engine.update_fields(dom.fi, dom.rho, dom.u, dom.flags, self.t, self.fx, self.fy, self.fz, dom.F)
            self.t_last_update_fields = self.t

    def read_from_device(self):
        if self.initialized:
            self.update_fields()
        self._gather()

    def update_force_field(self):
        for dom in self.domains:
            dom.engine.update_force_field(dom.fi, dom.flags, self.t, dom.F)

    # ---- FluidX3D-master/src/lbm.cpp:266-303, 1509-1554: per-domain kernel sums, added in float on the host
    def _object_sum(self, what, marker, center=(0.0, 0.0, 0.0)):
        total = np.zeros(3, dtype=np.float32)
        cells = 0
        for dom in self.domains:
            s = dom.engine.object_sum(what, dom.F, dom.flags, marker, center)
            total = (total + s[:3]).astype(np.float32)
            cells += int(s[3:4].view(np.uint32)[0])
        return total, cells

    def object_center_of_mass(self, marker=TYPE_S):
        if not self.initialized:
            self._scatter()
        total, cells = self._object_sum(0, marker)
        return (total / np.float32(cells)).astype(np.float32)

    def object_force(self, marker=TYPE_S):
        self.update_force_field()
        return self._object_sum(1, marker)[0]

    def object_torque(self, center, marker=TYPE_S):
        self.update_force_field()
        return self._object_sum(2, marker, center)[0]
