"""Python mirror of the reference's LBM host class over the C ABI of liblbmhip.so.

Mirrors FluidX3D-master/src/lbm.hpp:208-611 / lbm.cpp:1038-1124, 1354-1500, 2120-2182: a global lattice split
into Dx*Dy*Dz domains with 1-cell halos, host fields rho/u/flags/F, initialize(), do_time_step(), run(),
update_fields(), read_from_device().  Only ctypes and numpy are used; all arithmetic happens in the HIP kernels.
"""
import ctypes as C
import os

import numpy as np

from . import decomposition
from . import mesh as meshlib

STORAGE = {"FP32": 0, "FP16S": 1, "FP16C": 2}
COLLISION = {"SRT": 0, "TRT": 1}
FEATURES = {"UPDATE_FIELDS": 1, "VOLUME_FORCE": 2, "MOVING_BOUNDARIES": 4, "EQUILIBRIUM_BOUNDARIES": 8,
            "FORCE_FIELD": 16, "SUBGRID": 32}
FIELD_FI, FIELD_RHO, FIELD_U, FIELD_FLAGS, FIELD_F = range(5)
HALO_FI, HALO_RHO_U_FLAGS, HALO_FLAGS, HALO_F = range(4)
BUF_SEND_P, BUF_SEND_M, BUF_RECV_P, BUF_RECV_M = range(4)
TYPE_S, TYPE_E, TYPE_T, TYPE_F, TYPE_I, TYPE_G, TYPE_X, TYPE_Y, TYPE_MS = 1, 2, 4, 8, 16, 32, 64, 128, 3

# LBMHIP_LIB names another build of the library (diagnostic: comparing two builds on the same GPU)
lib_path = os.environ.get("LBMHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblbmhip.so")


class LbmHipError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("Nx", C.c_uint32), ("Ny", C.c_uint32), ("Nz", C.c_uint32), ("Dx", C.c_uint32), ("Dy", C.c_uint32),
                ("Dz", C.c_uint32), ("Ox", C.c_int32), ("Oy", C.c_int32), ("Oz", C.c_int32), ("velocity_set", C.c_uint32),
                ("storage", C.c_uint32), ("collision", C.c_uint32), ("features", C.c_uint32), ("nu", C.c_float),
                ("device", C.c_int32)]


_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise LbmHipError("%s is missing: build it with `make -C cfd_amd/csrc` (or __graft_entry__.build())" % lib_path)
        L = C.CDLL(lib_path)
        L.lbmhip_last_error.restype = C.c_char_p
        L.lbmhip_version.restype = C.c_char_p
        L.lbmhip_bytes_per_cell_device.restype = C.c_uint32
        L.lbmhip_bandwidth_bytes_per_cell.restype = C.c_uint32
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise LbmHipError(lib().lbmhip_last_error().decode())


def feature_mask(features):
    m = 0
    for f in features:
        m |= FEATURES[f]
    return m


def device_count():
    n = C.c_int(0)
    check(lib().lbmhip_device_count(C.byref(n)))
    return n.value


def float_string(v):
    """A float as the reference prints it, d.dddddddd[E<exponent>] (utilities.hpp:2599-2630, 2745-2754): the digits are taken
    in float32 arithmetic (scale into [1, 10) by 10^32 ... 10^1, cut the first digit, rest * 10^8, round half up), so the
    last digits can differ from a correctly rounded conversion; file headers carry these strings."""
    f = np.float32
    x, sign, e = f(abs(v)), "-" if v < 0 else "", 0
    with np.errstate(over="ignore", under="ignore"):
        if x >= f(10.0):
            for p in (32, 16, 8, 4, 2, 1):
                if x >= f(float("1e%d" % p)):
                    x, e = f(x * f(float("1e-%d" % p))), e + p
        if f(0.0) < x <= f(1.0):
            for p in (32, 16, 8, 4, 2, 1):
                if x < f(float("1e-%d" % (p - 1))):
                    x, e = f(x * f(float("1e%d" % p))), e - p
        lead = int(x)
        rest = f(f(x - f(lead)) * f(1e8))
        dec = int(rest)
        if f(rest - f(dec)) >= f(0.5):
            dec += 1
            if dec >= 100000000:
                dec, lead = 0, lead + 1
                if lead >= 10:
                    lead, e = 1, e + 1
    return "%s%d.%08d%s" % (sign, lead, dec, "E%d" % e if e else "")


class Domain:
    """One lbmhip_domain handle plus its host mirrors (the reference's LBM_Domain, lbm.hpp:20-204)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.h = C.c_void_p()
        check(lib().lbmhip_domain_create(C.byref(cfg), C.byref(self.h)))
        self.N = cfg.Nx * cfg.Ny * cfg.Nz
        self.has_F = bool(cfg.features & FEATURES["FORCE_FIELD"])
        self._host = {}

    # host mirrors are allocated on first use (a 1600^3 lattice has 70 GB of them; callers that keep data on the device
    # never touch them)
    def _mirror(self, name, count, dtype, fill):
        if name not in self._host:
            self._host[name] = np.full(count, fill, dtype=dtype)
        return self._host[name]

    @property
    def rho(self): return self._mirror("rho", self.N, np.float32, 1.0)
    @property
    def u(self): return self._mirror("u", 3 * self.N, np.float32, 0.0)
    @property
    def flags(self): return self._mirror("flags", self.N, np.uint8, 0)
    @property
    def F(self): return self._mirror("F", 3 * self.N, np.float32, 0.0) if self.has_F else None

    def close(self):
        if self.h:
            lib().lbmhip_domain_destroy(self.h)
            self.h = C.c_void_p()

    def upload(self, field, a, offset=0):
        check(lib().lbmhip_upload(self.h, field, a.ctypes.data_as(C.c_void_p), C.c_uint64(offset), C.c_uint64(a.nbytes)))

    def download(self, field, a, offset=0):
        check(lib().lbmhip_download(self.h, field, a.ctypes.data_as(C.c_void_p), C.c_uint64(offset), C.c_uint64(a.nbytes)))

    def download_fi(self):
        dt = np.float32 if self.cfg.storage == 0 else np.uint16
        a = np.empty(self.N * self.cfg.velocity_set, dtype=dt)
        self.download(FIELD_FI, a)
        return a

    def field_pointer(self, field):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib().lbmhip_field_pointer(self.h, field, C.byref(p), C.byref(n)))
        return p.value, n.value


class LBM:
    """LBM(N=(Nx,Ny,Nz), nu, D=(Dx,Dy,Dz), Q, storage, collision, features, f=(fx,fy,fz), devices=[...])

    Global host fields (numpy, [z,y,x]; u and F [component,z,y,x]) are filled by the caller before run().
    devices lists one HIP ordinal per domain; with fewer GPUs than domains every domain runs on device 0, which is
    how the reference tests its decomposition on one device (FluidX3D-master/src/lbm.cpp:949-953, 985-989).
    """

    def __init__(self, N, nu, D=(1, 1, 1), Q=19, storage="FP32", collision="SRT", features=(), f=(0.0, 0.0, 0.0), devices=None, overlap=True):
        L = lib()
        D = tuple(int(v) for v in D)
        if D[0] * D[1] * D[2] == 0:
            raise LbmHipError("You specified 0 LBM grid domains (%dx%dx%d)." % D)
        self.N = decomposition.rounded_size(N, D)  # lbm.cpp:1043-1054
        self.D, self.Q, self.storage, self.collision, self.features, self.nu = D, Q, storage, collision, tuple(features), float(nu)
        self.fx, self.fy, self.fz = (float(v) for v in f)
        mask = feature_mask(features)
        if not (mask & FEATURES["VOLUME_FORCE"]) and (self.fx != 0.0 or self.fy != 0.0 or self.fz != 0.0):
            raise LbmHipError("Volume force is set in LBM constructor, but VOLUME_FORCE is not enabled.")  # lbm.cpp:1211-1215
        self.H = decomposition.halo(D)
        self.local = decomposition.local_size(self.N, D)
        nd = D[0] * D[1] * D[2]
        if devices is None:
            have = device_count()
            devices = list(range(nd)) if have >= nd else [0] * nd
        self.domains = []
        try:
            for d in range(nd):
                off = decomposition.domain_offset(d, self.N, D)
                cfg = Config(self.local[0], self.local[1], self.local[2], D[0], D[1], D[2], off[0], off[1], off[2],
                             Q, STORAGE[storage], COLLISION[collision], mask, self.nu, devices[d])
                self.domains.append(Domain(cfg))
        except Exception:
            self.close()
            raise
        self._handles = (C.c_void_p * nd)(*[dom.h for dom in self.domains])
        self._packed_t = None  # time step of DDF faces that are packed but not yet exchanged (_do_two_time_steps)
        Nx, Ny, Nz = self.N
        self.rho = np.ones((Nz, Ny, Nx), dtype=np.float32)
        self.u = np.zeros((3, Nz, Ny, Nx), dtype=np.float32)
        self.flags = np.zeros((Nz, Ny, Nx), dtype=np.uint8)
        self.has_F = bool(mask & FEATURES["FORCE_FIELD"])
        self.F = np.zeros((3, Nz, Ny, Nx), dtype=np.float32) if self.has_F else None
        self.update_fields_ext = bool(mask & FEATURES["UPDATE_FIELDS"])
        self.t = 0
        self.initialized = False
        self.t_last_update_fields = None
        self.t_last_force_field = None
        # compute/communication overlap is used for pure z-slab decompositions
        self.overlap = bool(overlap) and D[0] == 1 and D[1] == 1 and D[2] > 1 and self.local[2] >= 5
        self._L = L

    # ---- life cycle
    def close(self):
        for dom in getattr(self, "domains", []):
            dom.close()
        self.domains = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- getters of the reference's class (lbm.hpp:454-472)
    def get_Nx(self): return self.N[0]
    def get_Ny(self): return self.N[1]
    def get_Nz(self): return self.N[2]
    def get_N(self): return self.N[0] * self.N[1] * self.N[2]
    def get_D(self): return self.D[0] * self.D[1] * self.D[2]
    def get_t(self): return self.t
    def get_nu(self): return self.nu
    def get_tau(self): return float(np.float32(3.0) * np.float32(self.nu) + np.float32(0.5))

    def set_f(self, fx, fy, fz):
        self.fx, self.fy, self.fz = float(fx), float(fy), float(fz)

    def bandwidth_bytes_per_cell(self):
        return self._L.lbmhip_bandwidth_bytes_per_cell(self.Q, STORAGE[self.storage], feature_mask(self.features))

    def bytes_per_cell_device(self):
        return self._L.lbmhip_bytes_per_cell_device(self.Q, STORAGE[self.storage], feature_mask(self.features))

    # ---- Memory_Container::write_device_to_vtk (lbm.hpp:298-340): legacy VTK, binary, big-endian, STRUCTURED_POINTS,
    # components interleaved per cell, file name <field>-<t as 9 digits>.vtk; lattice units (the SI factors of units.hpp are 1)
    def write_device_to_vtk(self, name, path=""):
        self.read_from_device()
        return self.write_host_to_vtk(name, path)

    def write_host_to_vtk(self, name, path="", units=(1.0, 1.0, 1.0)):
        """Legacy binary VTK like Memory_Container::write_vtk (lbm.hpp:298-340).  units = (m, kg, s) of one lattice unit
        (units.hpp:9-13): spacing, origin and the values of rho, u and F are scaled to SI like the reference does."""
        a = getattr(self, name)
        if a is None:
            raise LbmHipError("field %s does not exist (FORCE_FIELD is off)" % name)
        Nx, Ny, Nz = self.N
        d = 3 if a.ndim == 4 else 1
        if path == "" or path.endswith("/"):
            path = (path or "./") + "%s-%09d.vtk" % (name, self.t)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        kind = "unsigned_char" if a.dtype == np.uint8 else "float"
        m, kg, sec = (float(v) for v in units)
        f6 = float_string
        header = ("# vtk DataFile Version 3.0\nFluidX3D %s\nBINARY\nDATASET STRUCTURED_POINTS\nDIMENSIONS %d %d %d\nORIGIN %s %s %s\n"
                  "SPACING %s %s %s\nPOINT_DATA %d\nSCALARS data %s %d\nLOOKUP_TABLE default\n") % (
            os.path.basename(path), Nx, Ny, Nz, f6(m * (0.5 - 0.5 * Nx)), f6(m * (0.5 - 0.5 * Ny)), f6(m * (0.5 - 0.5 * Nz)), f6(m), f6(m), f6(m), Nx * Ny * Nz, kind, d)
        factor = np.float32({"rho": kg / m ** 3, "u": m / sec, "F": kg * m / sec ** 2}.get(name, 1.0))
        with open(path, "wb") as f:
            f.write(header.encode())
            for z in range(Nz):  # plane by plane: SoA -> interleaved, byte-swapped
                plane = np.moveaxis(a[:, z], 0, -1) if d == 3 else a[z]
                plane = np.ascontiguousarray(plane)
                f.write((factor * plane).astype(">f4").tobytes() if a.dtype == np.float32 else plane.astype(np.uint8).tobytes())
        return path

    def get_tuning(self, key):
        """Launch setting of the first domain ("cells_per_thread", "cells_together", "blocks_per_cu", "block", "autotune", "tuned")."""
        v = C.c_int()
        check(self._L.lbmhip_get_tuning(self.domains[0].h, key.encode(), C.byref(v)))
        return v.value

    def set_tuning(self, key, value):
        for dom in self.domains:
            check(self._L.lbmhip_set_tuning(dom.h, key.encode(), int(value)))

    # ---- global <-> per-domain mapping (Memory_Container::reference, lbm.hpp:265-288)
    def _slices(self, d):
        return decomposition.slices(d, self.N, self.D)

    def _view(self, a, comps):
        Lx, Ly, Lz = self.local
        return a.reshape((comps, Lz, Ly, Lx)) if comps > 1 else a.reshape((Lz, Ly, Lx))

    def write_to_device(self, fields=("rho", "u", "flags", "F")):
        for d, dom in enumerate(self.domains):
            g, l = self._slices(d)
            al = (slice(None),)
            if "rho" in fields:
                self._view(dom.rho, 1)[l] = self.rho[g]; dom.upload(FIELD_RHO, dom.rho)
            if "u" in fields:
                self._view(dom.u, 3)[al + l] = self.u[al + g]; dom.upload(FIELD_U, dom.u)
            if "flags" in fields:
                self._view(dom.flags, 1)[l] = self.flags[g]; dom.upload(FIELD_FLAGS, dom.flags)
            if "F" in fields and self.has_F:
                self._view(dom.F, 3)[al + l] = self.F[al + g]; dom.upload(FIELD_F, dom.F)

    def read_from_device(self, fields=("rho", "u", "flags", "F")):
        """Memory_Container::read_from_device (lbm.hpp:390-396): refresh rho/u on the device first if needed."""
        if self.initialized:
            self.update_fields()
        for d, dom in enumerate(self.domains):
            g, l = self._slices(d)
            al = (slice(None),)
            if "rho" in fields:
                dom.download(FIELD_RHO, dom.rho); self.rho[g] = self._view(dom.rho, 1)[l]
            if "u" in fields:
                dom.download(FIELD_U, dom.u); self.u[al + g] = self._view(dom.u, 3)[al + l]
            if "flags" in fields:
                dom.download(FIELD_FLAGS, dom.flags); self.flags[g] = self._view(dom.flags, 1)[l]
            if "F" in fields and self.has_F:
                dom.download(FIELD_F, dom.F); self.F[al + g] = self._view(dom.F, 3)[al + l]

    # ---- halo exchange between the domains of this process (LBM::communicate_field, lbm.cpp:2120-2182)
    def _communicate(self, what):
        L = self._L
        for axis in range(3):
            if self.D[axis] <= 1:
                continue
            for dom in self.domains:
                check(L.lbmhip_halo_pack(dom.h, what, axis, C.c_uint64(self.t)))
            check(L.lbmhip_halo_exchange_local(self._handles, len(self.domains), what, axis))
            for dom in self.domains:
                check(L.lbmhip_halo_unpack(dom.h, what, axis, C.c_uint64(self.t)))

    def sync(self):
        for dom in self.domains:
            check(self._L.lbmhip_sync(dom.h))

    # ---- LBM::initialize (lbm.cpp:1354-1409)
    def initialize(self):
        L = self._L
        self.write_to_device()
        if self.has_F:
            self._communicate(HALO_F)
        self.t = 1  # the communicate calls at initialization need an odd time step
        self._communicate(HALO_RHO_U_FLAGS)
        for dom in self.domains:
            check(L.lbmhip_initialize(dom.h))
        self._communicate(HALO_RHO_U_FLAGS)
        self._communicate(HALO_FI)
        self.sync()
        self.t = 0
        self.initialized = True
        self.t_last_update_fields = 0 if self.update_fields_ext else None
        self.t_last_force_field = None

    # ---- LBM::do_time_step (lbm.cpp:1411-1459) without the per-step host synchronisation
    def do_time_step(self):
        L = self._L
        t, fx, fy, fz = C.c_uint64(self.t), C.c_float(self.fx), C.c_float(self.fy), C.c_float(self.fz)
        if self.overlap:
            # planes next to the halos first, their exchange on the communication streams while the rest collides
            Lz = self.local[2]
            for dom in self.domains:
                check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, 1, 2))
                check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, Lz - 2, Lz - 1))
            for dom in self.domains:
                check(L.lbmhip_comm_begin(dom.h))
            self._communicate(HALO_FI)
            for dom in self.domains:
                check(L.lbmhip_comm_end(dom.h))
            for dom in self.domains:
                check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, 2, Lz - 2))
            for dom in self.domains:
                check(L.lbmhip_comm_join(dom.h))
        else:
            for dom in self.domains:
                check(L.lbmhip_stream_collide(dom.h, t, fx, fy, fz))
            self._communicate(HALO_FI)
        self.t += 1
        if self.update_fields_ext:
            self.t_last_update_fields = self.t

    def _pairs_slab_steps(self):
        """z-slabs whose engine runs the inner planes of two steps in one launch (lbmhip_get_tuning "pairs_steps")."""
        if not (self.overlap and self.D[0] == 1 and self.D[1] == 1 and self.D[2] > 1 and self.local[2] >= 12):
            return False
        v = C.c_int(0)
        for dom in self.domains:
            check(self._L.lbmhip_get_tuning(dom.h, b"pairs_steps", C.byref(v)))
            if not v.value:
                return False
        return True

    def _finish_packed_exchange(self):
        """The exchange of the DDF faces _do_two_time_steps left packed (on whichever streams are current)."""
        if self._packed_t is None:
            return
        t, self._packed_t = self._packed_t, None
        check(self._L.lbmhip_halo_exchange_local(self._handles, len(self.domains), HALO_FI, 2))
        for dom in self.domains:
            check(self._L.lbmhip_halo_unpack(dom.h, HALO_FI, 2, C.c_uint64(t)))

    def _flush_packed_exchange(self):
        if self._packed_t is None:
            return
        L = self._L
        for dom in self.domains:
            check(L.lbmhip_comm_begin(dom.h))
        self._finish_packed_exchange()
        for dom in self.domains:
            check(L.lbmhip_comm_end(dom.h))
        for dom in self.domains:
            check(L.lbmhip_comm_join(dom.h))

    def _do_two_time_steps(self):
        """Two steps of z-slabs with the inner planes of both in one launch per domain (lbmhip_stream_collide_pair_planes), both
        exchanges beside an inner launch: on the communication streams the exchange the call before left packed, step t next to the
        halos and its exchange; beside them the inner launch; after the join step t+1 on the six planes near the halos, whose faces
        are packed and left for the next call (or for run(), after the last pair).  Same bits as two calls of do_time_step (a cell
        of step t+1 in plane p depends on step t in the planes p-1 .. p+1 only, and step t+2 on the planes 2 .. Lz-3 owns no slot that
        the unpack of step t+1 writes)."""
        L = self._L
        Lz = self.local[2]
        fx, fy, fz = C.c_float(self.fx), C.c_float(self.fy), C.c_float(self.fz)
        t = C.c_uint64(self.t)
        for dom in self.domains:
            check(L.lbmhip_comm_begin(dom.h))
        self._finish_packed_exchange()
        for dom in self.domains:
            check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, 1, 2))
            check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, Lz - 2, Lz - 1))
        self._communicate(HALO_FI)
        for dom in self.domains:
            check(L.lbmhip_comm_end(dom.h))
        for dom in self.domains:
            check(L.lbmhip_stream_collide_pair_planes(dom.h, t, fx, fy, fz, 2, Lz - 2))  # step t on 2 .. Lz-3, step t+1 on 4 .. Lz-5
        for dom in self.domains:
            check(L.lbmhip_comm_join(dom.h))
        self.t += 1
        t = C.c_uint64(self.t)
        for dom in self.domains:
            check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, 1, 4))
            check(L.lbmhip_stream_collide_planes(dom.h, t, fx, fy, fz, Lz - 4, Lz - 1))
        for dom in self.domains:
            check(L.lbmhip_halo_pack(dom.h, HALO_FI, 2, t))
        self._packed_t = self.t
        self.t += 1
        if self.update_fields_ext:
            self.t_last_update_fields = self.t

    def run(self, steps):
        if not self.initialized:
            self.initialize()
        if int(steps) >= 2 and self._pairs_slab_steps():
            for _ in range(int(steps) // 2):
                self._do_two_time_steps()
            self._flush_packed_exchange()
            steps = int(steps) % 2
        if len(self.domains) == 1 and int(steps) > 0:  # no exchange between the steps: the library may run them two per launch
            check(self._L.lbmhip_stream_collide_steps(self.domains[0].h, C.c_uint64(self.t), C.c_uint32(int(steps)), C.c_float(self.fx), C.c_float(self.fy), C.c_float(self.fz)))
            self.t += int(steps)
            if self.update_fields_ext:
                self.t_last_update_fields = self.t
        else:
            for _ in range(int(steps)):
                self.do_time_step()
        self.sync()

    # ---- LBM::update_fields / enqueue_update_fields (lbm.cpp:238-246, 1491-1496)
    def update_fields(self):
        if not self.update_fields_ext and self.t != self.t_last_update_fields:
            for dom in self.domains:
                check(self._L.lbmhip_update_fields(dom.h, C.c_uint64(self.t), C.c_float(self.fx), C.c_float(self.fy), C.c_float(self.fz)))
            self.t_last_update_fields = self.t
        self.sync()

    # ---- LBM::update_force_field (lbm.cpp:258-265, 1503-1508)
    def update_force_field(self):
        if self.t != self.t_last_force_field:
            for dom in self.domains:
                check(self._L.lbmhip_update_force_field(dom.h, C.c_uint64(self.t)))
            self.t_last_force_field = self.t
        self.sync()

    # ---- LBM::object_center_of_mass / object_force / object_torque (lbm.cpp:1509-1554)
    def _object_sum(self, what, flag_marker, center=(0.0, 0.0, 0.0)):
        size = (C.c_uint32 * 3)(*self.N)
        c = (C.c_float * 3)(*[float(v) for v in center])
        total = np.zeros(4, dtype=np.float64)
        for dom in self.domains:
            s = (C.c_double * 4)()
            check(self._L.lbmhip_object_sum(dom.h, what, C.c_uint8(flag_marker), c, size, s))
            total += np.array(s[:])
        return total

    def object_center_of_mass(self, flag_marker=TYPE_S):
        if not self.initialized:
            self.write_to_device(("flags",))
        s = self._object_sum(0, flag_marker)
        return tuple(s[:3] / s[3]) if s[3] > 0 else (0.0, 0.0, 0.0)

    def object_force(self, flag_marker=TYPE_S):
        self.update_force_field()
        return tuple(self._object_sum(1, flag_marker)[:3])

    def object_torque(self, rotation_center, flag_marker=TYPE_S):
        self.update_force_field()
        return tuple(self._object_sum(2, flag_marker, rotation_center)[:3])

    def reset(self):
        self.initialized = False

    # ---- LBM::voxelize_mesh_on_device / voxelize_stl (lbm.cpp:1645-1670, 1728-1752)
    def voxelize_mesh_on_device(self, mesh, flag=TYPE_S, rotation_center=(0, 0, 0), linear_velocity=(0, 0, 0), rotational_velocity=(0, 0, 0), static=False):
        """static=True stages the triangles in the library's fixed device buffer (lbmhip_voxelize_mesh_static: no device allocation)."""
        direction, bbox, rc, lin, rot = meshlib.voxelize_parameters(mesh, rotation_center, linear_velocity, rotational_velocity)
        call = self._L.lbmhip_voxelize_mesh_static if static else self._L.lbmhip_voxelize_mesh
        fp = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.c_void_p)
        p0, p1, p2 = (np.ascontiguousarray(p, dtype=np.float32) for p in (mesh.p0, mesh.p1, mesh.p2))
        for dom in self.domains:
            check(call(dom.h, C.c_uint32(direction), C.c_uint64(self.t + 1), C.c_uint8(flag), fp(p0), fp(p1), fp(p2),
                       C.c_uint32(mesh.triangle_number), fp(bbox), fp(rc), fp(lin), fp(rot)))
        if not self.initialized:  # the host copies follow the device until the first run() (lbm.cpp:1666-1669)
            self._read_raw(("flags", "u"))

    def unvoxelize_mesh_on_device(self, mesh, flag=TYPE_S):
        bbox = np.concatenate([mesh.pmin, mesh.pmax]).astype(np.float32)
        for dom in self.domains:
            check(self._L.lbmhip_unvoxelize_mesh(dom.h, C.c_uint8(flag), bbox.ctypes.data_as(C.c_void_p)))
        self.sync()

    def voxelize_stl(self, path, center=None, rotation=None, size=0.0, flag=TYPE_S):
        c = self.center() if center is None else center
        m = meshlib.read_stl(path, self.size(), c, rotation, size)
        self.write_to_device(("flags",))
        self.voxelize_mesh_on_device(m, flag)
        self._read_raw(("flags",))
        return m

    def _read_raw(self, fields):
        """Device -> host copies without the update_fields refresh (used before the simulation is initialized)."""
        was, self.initialized = self.initialized, False
        try:
            self.read_from_device(fields)
        finally:
            self.initialized = was

    def size(self):
        return tuple(float(v) for v in self.N)

    def center(self):
        return tuple(0.5 * float(v) - 0.5 for v in self.N)
