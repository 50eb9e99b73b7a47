"""One process per GPU: z-slab decomposition with the per-step DDF halo exchange over torch.distributed.

This is the multi-process counterpart of LBM::communicate_field (FluidX3D-master/src/lbm.cpp:2120-2182): the
reference swaps host pointers between domains that live in one process; here every rank owns one slab
(Dx=Dy=1, Dz=world) and sends its two face buffers to its z-neighbours (RCCL send/recv over xGMI when the
backend is "nccl", which is RCCL on ROCm).  There is no collective on the data path, only point-to-point
exchanges with at most two peers per rank; the periodic wrap connects the last rank with rank 0.

D=(Dx, Dy, Dz) with Dx*Dy*Dz ranks splits the lattice into blocks as the reference does (lbm.cpp:1038-1073, rank =
x+(y+z*Dy)*Dx): the exchange then runs axis after axis, x, y, z, each face including the halo cells of the axes
before it, so edges and corners arrive in two or three hops (lbm.cpp:2162-2181).  Blocks step in the reference's own
order (whole domain, then the exchange); the overlapped and the two-steps-per-launch schedules below are for slabs.

Per time step (overlapped order):
  1. the communication stream waits for the previous step
  2. stream_collide on the two lattice planes next to the halos, pack -> send/recv -> unpack   (communication stream)
  3. stream_collide on all other planes, concurrently with 2                                   (compute stream)
  4. the compute stream waits for 2
Every DDF slot is read and written by exactly one cell per step, so the two boundary planes may collide while the
interior does; the slots that pack reads / unpack writes belong to cells of the two boundary planes (pack) or are only
read by them in the NEXT step (unpack), so 2 and 3 never touch the same memory.

The slab engine is abstract (SlabOps) so that the exchange logic can be driven by CPU tensors in tests; the
product implementation is HipSlabOps (HIP kernels through the C ABI, staging buffers owned by torch).
"""
import contextlib
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import decomposition
from . import lbm as L

WHAT = {"fi": L.HALO_FI, "rho_u_flags": L.HALO_RHO_U_FLAGS, "flags": L.HALO_FLAGS, "F": L.HALO_F}


class SlabOps:
    """What DistributedLBM needs from the engine that owns this rank's slab."""
    local = (0, 0, 0)          # local lattice size including the halo layers of every split axis
    buffers = None             # dict send_p, send_m, recv_p, recv_m -> 1-D uint8 torch tensors
    def upload(self, rho, u, flags, F): raise NotImplementedError
    def upload_plane(self, name, z_local, a): raise NotImplementedError
    def download(self, names): raise NotImplementedError
    def initialize(self): raise NotImplementedError
    def collide_planes(self, t, f, z0, z1): raise NotImplementedError
    def pairs_steps(self): return True  # does collide_pair_planes gain anything over two collide_planes calls? (decides the schedule of run())
    def collide_pair_planes(self, t, f, z0, z1):  # step t on [z0, z1) and step t+1 on [z0+2, z1-2), which depends on nothing else
        self.collide_planes(t, f, z0, z1)
        self.collide_planes(t + 1, f, z0 + 2, z1 - 2)
    def collide_steps(self, t, f, steps):  # `steps` whole-lattice steps with nothing in between (a single rank only)
        for k in range(steps):
            self.collide_planes(t + k, f, 0, self.local[2])
    def update_fields(self, t, f): raise NotImplementedError
    def update_force_field(self, t): raise NotImplementedError
    def pack(self, what, t, axis=2): raise NotImplementedError
    def unpack(self, what, t, axis=2): raise NotImplementedError
    def halo_bytes(self, what, axis=2): raise NotImplementedError
    def scope(self): return contextlib.nullcontext()  # makes the engine's compute stream the current one for the block
    def fork(self): pass       # communication stream waits for the compute stream and becomes current
    def end_fork(self): pass   # back to the compute stream
    def join(self): pass       # compute stream waits for the communication stream
    def sync(self): pass


class HipSlabOps(SlabOps):
    """The slab lives in an lbmhip domain on this rank's GPU; torch owns the staging buffers and the streams."""

    def __init__(self, local, world, Q, storage, collision, features, nu, offset, device):
        """world: number of z-slabs, or (Dx, Dy, Dz); offset: global z (or (x, y, z)) of the first local cell, halo included."""
        lib = L.lib()
        self.lib = lib
        self.device = torch.device("cuda", device)
        mask = L.feature_mask(features)
        D = (1, 1, int(world)) if np.isscalar(world) else tuple(int(v) for v in world)
        offset = (0, 0, int(offset)) if np.isscalar(offset) else tuple(int(v) for v in offset)
        cfg = L.Config(local[0], local[1], local[2], D[0], D[1], D[2], offset[0], offset[1], offset[2], Q, L.STORAGE[storage], L.COLLISION[collision], mask, float(nu), device)
        self.dom = L.Domain(cfg)
        self.local = tuple(local)
        self.has_F = bool(mask & L.FEATURES["FORCE_FIELD"])
        self.compute = torch.cuda.Stream(device=self.device)
        self.comm = torch.cuda.Stream(device=self.device)
        L.check(lib.lbmhip_set_stream(self.dom.h, C.c_void_p(self.compute.cuda_stream)))
        self.buffers = None
        if D[0] * D[1] * D[2] > 1:
            face = max(local[(a + 1) % 3] * local[(a + 2) % 3] for a in range(3) if D[a] > 1)  # cells of the largest face that is exchanged
            cap = face * max(decomposition.halo_bytes_per_cell("fi", Q, 4 if storage == "FP32" else 2), 17)
            self.buffers = {k: torch.empty(cap, dtype=torch.uint8, device=self.device) for k in ("send_p", "send_m", "recv_p", "recv_m")}
            b = self.buffers
            L.check(lib.lbmhip_halo_set_buffers(self.dom.h, C.c_void_p(b["send_p"].data_ptr()), C.c_void_p(b["send_m"].data_ptr()),
                                                C.c_void_p(b["recv_p"].data_ptr()), C.c_void_p(b["recv_m"].data_ptr()), C.c_uint64(cap)))
        self._stream_ctx = None

    def close(self):
        self.dom.close()

    def upload(self, rho, u, flags, F):
        d = self.dom
        d.rho[...] = rho.ravel(); d.u[...] = u.ravel(); d.flags[...] = flags.ravel()
        d.upload(L.FIELD_RHO, d.rho); d.upload(L.FIELD_U, d.u); d.upload(L.FIELD_FLAGS, d.flags)
        if self.has_F:
            d.F[...] = F.ravel(); d.upload(L.FIELD_F, d.F)

    def upload_plane(self, name, z_local, a):
        """One lattice plane ([y, x]; u and F: [3, y, x]) into the slab, without a host copy of the whole slab."""
        d = self.dom
        Lx, Ly, _ = self.local
        field, item = {"rho": (L.FIELD_RHO, 4), "u": (L.FIELD_U, 4), "flags": (L.FIELD_FLAGS, 1), "F": (L.FIELD_F, 4)}[name]
        a = np.ascontiguousarray(a, dtype=np.uint8 if name == "flags" else np.float32)
        if name in ("u", "F"):
            for c in range(3):
                d.upload(field, a[c], offset=(c * d.N + z_local * Lx * Ly) * item)
        else:
            d.upload(field, a, offset=z_local * Lx * Ly * item)

    def download(self, names):
        d = self.dom
        out = {}
        for name in names:
            if name == "rho": d.download(L.FIELD_RHO, d.rho); out[name] = d.rho
            elif name == "u": d.download(L.FIELD_U, d.u); out[name] = d.u
            elif name == "flags": d.download(L.FIELD_FLAGS, d.flags); out[name] = d.flags
            elif name == "F" and self.has_F: d.download(L.FIELD_F, d.F); out[name] = d.F
            elif name == "fi": out[name] = d.download_fi()
        return out

    def initialize(self):
        L.check(self.lib.lbmhip_initialize(self.dom.h))

    def collide_planes(self, t, f, z0, z1):
        L.check(self.lib.lbmhip_stream_collide_planes(self.dom.h, C.c_uint64(t), C.c_float(f[0]), C.c_float(f[1]), C.c_float(f[2]), int(z0), int(z1)))

    def pairs_steps(self):
        v = C.c_int(0)
        L.check(self.lib.lbmhip_get_tuning(self.dom.h, b"pairs_steps", C.byref(v)))
        return bool(v.value)

    def collide_pair_planes(self, t, f, z0, z1):
        L.check(self.lib.lbmhip_stream_collide_pair_planes(self.dom.h, C.c_uint64(t), C.c_float(f[0]), C.c_float(f[1]), C.c_float(f[2]), int(z0), int(z1)))

    def collide_steps(self, t, f, steps):
        L.check(self.lib.lbmhip_stream_collide_steps(self.dom.h, C.c_uint64(t), C.c_uint32(steps), C.c_float(f[0]), C.c_float(f[1]), C.c_float(f[2])))

    def update_fields(self, t, f):
        L.check(self.lib.lbmhip_update_fields(self.dom.h, C.c_uint64(t), C.c_float(f[0]), C.c_float(f[1]), C.c_float(f[2])))

    def update_force_field(self, t):
        L.check(self.lib.lbmhip_update_force_field(self.dom.h, C.c_uint64(t)))

    def pack(self, what, t, axis=2):
        L.check(self.lib.lbmhip_halo_pack(self.dom.h, WHAT[what], int(axis), C.c_uint64(t)))

    def unpack(self, what, t, axis=2):
        L.check(self.lib.lbmhip_halo_unpack(self.dom.h, WHAT[what], int(axis), C.c_uint64(t)))

    def halo_bytes(self, what, axis=2):
        n = C.c_uint64()
        L.check(self.lib.lbmhip_halo_bytes(self.dom.h, WHAT[what], int(axis), C.byref(n)))
        return n.value

    def scope(self):
        # torch.distributed orders its transfers against torch's CURRENT stream, so the stream the kernels are launched
        # on has to be the current one whenever an exchange is issued
        return torch.cuda.stream(self.compute)

    def fork(self):
        self.comm.wait_stream(self.compute)
        self._stream_ctx = torch.cuda.stream(self.comm)
        self._stream_ctx.__enter__()
        L.check(self.lib.lbmhip_set_stream(self.dom.h, C.c_void_p(self.comm.cuda_stream)))

    def end_fork(self):
        L.check(self.lib.lbmhip_set_stream(self.dom.h, C.c_void_p(self.compute.cuda_stream)))
        self._stream_ctx.__exit__(None, None, None)
        self._stream_ctx = None

    def join(self):
        self.compute.wait_stream(self.comm)

    def sync(self):
        self.compute.synchronize()
        self.comm.synchronize()
        L.check(self.lib.lbmhip_sync(self.dom.h))  # also the library's own look at what its two-step launches reported


class DistributedLBM:
    """This rank's part of a global Nx x Ny x Nz lattice: one of `world` z-slabs, or with D=(Dx, Dy, Dz) one of Dx*Dy*Dz blocks.

    Host fields rho/u/flags(/F) cover the rank's own cells only ([z, y, x] local; `origin` is their global offset),
    so setups written against N/origin/rho.shape fill them without materialising the global lattice.
    host_fields=False keeps no host copy at all (slabs of 10^9 cells): the part starts as the engine creates it (rho = 1,
    everything else 0), fill_planes() writes the planes that differ, and read_from_device() is not available.
    """

    def __init__(self, N, nu, Q=19, storage="FP32", collision="SRT", features=(), f=(0.0, 0.0, 0.0), rank=None, world=None, ops_factory=None, overlap=True, group=None, host_fields=True, D=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        D = (1, 1, self.world) if D is None else tuple(int(v) for v in D)
        if D[0] * D[1] * D[2] != self.world:
            raise ValueError("D = %r needs %d ranks, the group has %d" % (D, D[0] * D[1] * D[2], self.world))
        self.N = decomposition.rounded_size(N, D)
        self.D, self.Q, self.storage, self.collision, self.features, self.nu = D, Q, storage, collision, tuple(features), float(nu)
        self.f = tuple(float(v) for v in f)
        self.local = decomposition.local_size(self.N, D)
        self.n = tuple(self.N[k] // D[k] for k in range(3))       # own cells per axis
        self.nz = self.n[2]
        self.halo = decomposition.halo(D)                         # 1 on every split axis
        self.hz = self.halo[2]
        coords = decomposition.domain_coordinates(self.rank, D)
        self.origin = tuple(coords[k] * self.n[k] for k in range(3))
        self.slabs = D[0] == 1 and D[1] == 1
        # (next, previous) rank along every axis, periodic (lbm.cpp:2130-2132)
        self.neighbours = [[decomposition.neighbour(self.rank, D, a, 1), decomposition.neighbour(self.rank, D, a, -1)] for a in range(3)]
        if ops_factory is None:
            dev = torch.cuda.current_device()
            offset = tuple(self.origin[k] - self.halo[k] for k in range(3))
            ops_factory = lambda: HipSlabOps(self.local, D, Q, storage, collision, features, nu, offset, dev)
        self.ops = ops_factory()
        nx, ny, nz = self.n
        self.has_F = "FORCE_FIELD" in features
        self.host_fields = bool(host_fields)
        self.rho = self.u = self.flags = self.F = None
        if self.host_fields:
            self.rho = np.ones((nz, ny, nx), dtype=np.float32)
            self.u = np.zeros((3, nz, ny, nx), dtype=np.float32)
            self.flags = np.zeros((nz, ny, nx), dtype=np.uint8)
            self.F = np.zeros((3, nz, ny, nx), dtype=np.float32) if self.has_F else None
        self.update_fields_ext = "UPDATE_FIELDS" in features
        self.t = 0
        self.initialized = False
        self.t_last_update_fields = None
        self.overlap = bool(overlap) and self.world > 1 and self.slabs and self.local[2] >= 5
        self.pair_steps = True  # run() takes the steps of a slab two at a time (_do_two_time_steps)
        self._host_staging = None
        self._packed_t = None  # time step of DDF faces that are packed but not yet exchanged (_do_two_time_steps)

    # the z neighbours under their old names (a single rank that is its own neighbour sets both to 0)
    next = property(lambda self: self.neighbours[2][0], lambda self, r: self.neighbours[2].__setitem__(0, r))
    prev = property(lambda self: self.neighbours[2][1], lambda self, r: self.neighbours[2].__setitem__(1, r))

    def get_N(self):
        return self.N[0] * self.N[1] * self.N[2]

    def close(self):
        if hasattr(self.ops, "close"):
            self.ops.close()

    # ---- the exchange: + face to the next rank's - halo, - face to the previous rank's + halo (lbm.cpp:2162-2181)
    def _exchange(self, nbytes, axis=2):
        b = self.ops.buffers
        backend = dist.get_backend(self.group)
        views = {k: v[:nbytes] for k, v in b.items()}
        on_device = views["send_p"].is_cuda
        if on_device and backend != "nccl":  # rehearsal path (gloo): stage through host memory
            if self._host_staging is None or self._host_staging["send_p"].numel() < nbytes:
                self._host_staging = {k: torch.empty(b[k].numel(), dtype=torch.uint8).pin_memory() for k in b}
            h = {k: v[:nbytes] for k, v in self._host_staging.items()}
            h["send_p"].copy_(views["send_p"]); h["send_m"].copy_(views["send_m"])
            torch.cuda.current_stream().synchronize()
            self._p2p(h, axis)
            views["recv_p"].copy_(h["recv_p"]); views["recv_m"].copy_(h["recv_m"])
        else:
            self._p2p(views, axis)

    def _p2p(self, v, axis=2):
        g = self.group
        gr = (lambda r: dist.get_global_rank(g, r)) if g is not None else (lambda r: r)
        nxt, prv = self.neighbours[axis]
        ops = [dist.P2POp(dist.isend, v["send_p"], gr(nxt), g, tag=0), dist.P2POp(dist.isend, v["send_m"], gr(prv), g, tag=1),
               dist.P2POp(dist.irecv, v["recv_m"], gr(prv), g, tag=0), dist.P2POp(dist.irecv, v["recv_p"], gr(nxt), g, tag=1)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def _communicate(self, what):
        if self.world <= 1:
            return
        for axis in range(3):  # x, then y with the x halos in its faces, then z with both (lbm.cpp:2162-2181)
            if self.D[axis] > 1:
                self.ops.pack(what, self.t, axis)
                self._exchange(self.ops.halo_bytes(what, axis), axis)
                self.ops.unpack(what, self.t, axis)

    def _finish_packed_exchange(self):
        """The exchange whose faces _do_two_time_steps packed and left for later (on the communication stream of an overlapped run)."""
        if self._packed_t is not None:
            t, self._packed_t = self._packed_t, None
            self._exchange(self.ops.halo_bytes("fi"))
            self.ops.unpack("fi", t)

    # ---- LBM::initialize (lbm.cpp:1354-1409)
    def initialize(self):
        with self.ops.scope():
            self._initialize()

    def fill_planes(self, plane):
        """plane(z_global) -> {"flags": [y, x], "rho": [y, x], "u": [3, y, x], "F": [3, y, x]} (any subset, or None) for every
        plane this rank owns, written straight into the slab; the halo planes follow from the exchange in initialize()."""
        hx, hy, _ = self.halo
        (ox, oy, _), (nx, ny, _), (Lx, Ly, _) = self.origin, self.n, self.local
        for k in range(self.nz):
            got = plane(self.origin[2] + k) or {}
            for name, a in got.items():
                if not self.slabs:  # the rank's part of the global plane, inside a frame of halo cells that the exchange fills
                    a = np.asarray(a)
                    framed = np.full(a.shape[:-2] + (Ly, Lx), 1 if name == "rho" else 0, dtype=a.dtype)
                    framed[..., hy:hy + ny, hx:hx + nx] = a[..., oy:oy + ny, ox:ox + nx]
                    a = framed
                self.ops.upload_plane(name, k + self.hz, a)

    def _initialize(self):
        if self.host_fields:
            self._upload_host_fields()
        if self.has_F:
            self._communicate("F")
        self.t = 1
        self._communicate("rho_u_flags")
        self.ops.initialize()
        self._communicate("rho_u_flags")
        self._communicate("fi")
        self.ops.sync()
        self.t = 0
        self.initialized = True
        self.t_last_update_fields = 0 if self.update_fields_ext else None

    def _interior(self):
        """Index of the rank's own cells inside its local [z, y, x] arrays."""
        (hx, hy, hz), (nx, ny, nz) = self.halo, self.n
        return (slice(hz, hz + nz), slice(hy, hy + ny), slice(hx, hx + nx))

    def _upload_host_fields(self):
        Lx, Ly, Lz = self.local
        own = self._interior()
        rho = np.ones((Lz, Ly, Lx), dtype=np.float32); rho[own] = self.rho
        u = np.zeros((3, Lz, Ly, Lx), dtype=np.float32); u[(slice(None),) + own] = self.u
        flags = np.zeros((Lz, Ly, Lx), dtype=np.uint8); flags[own] = self.flags
        F = None
        if self.has_F:
            F = np.zeros((3, Lz, Ly, Lx), dtype=np.float32); F[(slice(None),) + own] = self.F
        self.ops.upload(rho, u, flags, F)

    # ---- LBM::do_time_step (lbm.cpp:1411-1459)
    def do_time_step(self):
        with self.ops.scope():
            self._do_time_step()

    def _do_time_step(self):
        Lz = self.local[2]
        o = self.ops
        if self.world <= 1:
            o.collide_planes(self.t, self.f, 0, Lz)
        elif self.overlap:
            o.fork()
            try:
                o.collide_planes(self.t, self.f, 1, 2)
                o.collide_planes(self.t, self.f, Lz - 2, Lz - 1)
                self._communicate("fi")
            finally:
                o.end_fork()
            o.collide_planes(self.t, self.f, 2, Lz - 2)
            o.join()
        else:
            o.collide_planes(self.t, self.f, 0, Lz)
            self._communicate("fi")
        self.t += 1
        if self.update_fields_ext:
            self.t_last_update_fields = self.t

    def _do_two_time_steps(self):
        """Steps t and t+1 of a slab, the inner planes of both in one launch (lbmhip_stream_collide_pair_planes), both exchanges hidden:
          [exchange t-1, packed by the call before]; step t on the planes next to the halos, exchange t      (communication stream)
          step t on the planes 2 .. Lz-3 and step t+1 on the planes 4 .. Lz-5                                (compute stream, beside them)
          step t+1 on the planes 1..3 and Lz-4..Lz-2, which need the halos of step t; pack the faces of t+1  (compute stream, after the join)
        The exchange of step t+1 is left to the next call (or to run(), after the last pair), where it runs beside the next inner
        launch: step t+2 on the planes 2 .. Lz-3 owns no slot in a halo plane and none that the unpack writes - those belong to the
        cells of the planes 1 and Lz-2, whose step t+2 follows the unpack on the communication stream.
        A cell of step t+1 in plane p depends on step t in the planes p-1 .. p+1 (the entry point keeps two planes of distance),
        so the planes 4 .. Lz-5 need nothing from a halo; every cell still sees exactly the values of the step before (same bits as
        two calls of _do_time_step)."""
        Lz = self.local[2]
        o = self.ops
        o.fork()
        try:
            self._finish_packed_exchange()
            o.collide_planes(self.t, self.f, 1, 2)
            o.collide_planes(self.t, self.f, Lz - 2, Lz - 1)
            self._communicate("fi")
        finally:
            o.end_fork()
        o.collide_pair_planes(self.t, self.f, 2, Lz - 2)
        o.join()
        self.t += 1
        o.collide_planes(self.t, self.f, 1, 4)
        o.collide_planes(self.t, self.f, Lz - 4, Lz - 1)
        o.pack("fi", self.t)
        self._packed_t = self.t
        self.t += 1
        if self.update_fields_ext:
            self.t_last_update_fields = self.t

    def _flush_packed_exchange(self):
        if self._packed_t is not None:
            o = self.ops
            o.fork()
            try:
                self._finish_packed_exchange()
            finally:
                o.end_fork()
            o.join()

    def run(self, steps):
        with self.ops.scope():
            if not self.initialized:
                self._initialize()
            # two steps at a time where the engine then runs the inner planes of both in one launch; otherwise every step keeps its exchange hidden
            if self.world > 1 and self.overlap and self.pair_steps and self.local[2] >= 12 and int(steps) >= 2 and self.ops.pairs_steps():
                for _ in range(int(steps) // 2):
                    self._do_two_time_steps()
                self._flush_packed_exchange()
                steps = int(steps) % 2
            if self.world <= 1 and int(steps) > 0:  # no exchange between the steps: the engine may pair them (lbmhip_stream_collide_steps)
                self.ops.collide_steps(self.t, self.f, int(steps))
                self.t += int(steps)
                if self.update_fields_ext:
                    self.t_last_update_fields = self.t
            else:
                for _ in range(int(steps)):
                    self._do_time_step()
        self.ops.sync()

    def update_fields(self):
        if not self.update_fields_ext and self.t != self.t_last_update_fields:
            self.ops.update_fields(self.t, self.f)
            self.t_last_update_fields = self.t
        self.ops.sync()

    def read_from_device(self):
        if not self.host_fields:
            raise RuntimeError("this DistributedLBM keeps no host fields (host_fields=False)")
        if self.initialized:
            self.update_fields()
        Lx, Ly, Lz = self.local
        own = self._interior()
        got = self.ops.download(("rho", "u", "flags") + (("F",) if self.has_F else ()))
        self.rho[...] = got["rho"].reshape(Lz, Ly, Lx)[own]
        self.u[...] = got["u"].reshape(3, Lz, Ly, Lx)[(slice(None),) + own]
        self.flags[...] = got["flags"].reshape(Lz, Ly, Lx)[own]
        if self.has_F:
            self.F[...] = got["F"].reshape(3, Lz, Ly, Lx)[(slice(None),) + own]
