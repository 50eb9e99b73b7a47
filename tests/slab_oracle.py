"""Test-only SlabOps over the CPU oracle, so the torch.distributed exchange logic of cfd_amd.distributed can run
on CPU ranks (gloo).  The product never imports this."""
import numpy as np
import torch

from cfd_amd.distributed import SlabOps
from oracle import harness as H


class OracleSlabOps(SlabOps):
    def __init__(self, local, world, Q, storage, collision, features, nu):
        """world: number of z-slabs, or (Dx, Dy, Dz)."""
        self.local = tuple(local)
        D = (1, 1, int(world)) if np.isscalar(world) else tuple(int(v) for v in world)
        self.D = D
        self.engine = H.PortEngine(local, D, Q, storage, collision, features, nu)
        n = local[0] * local[1] * local[2]
        self.n, self.Q, self.storage = n, Q, storage
        self.fi = np.zeros(Q * n, dtype=np.float32 if storage == "FP32" else np.uint16)
        self.rho = np.ones(n, dtype=np.float32)
        self.u = np.zeros(3 * n, dtype=np.float32)
        self.flags = np.zeros(n, dtype=np.uint8)
        self.F = np.zeros(3 * n, dtype=np.float32) if "FORCE_FIELD" in features else None
        self.per_cell = {"fi": H.TRANSFERS[Q] * (4 if storage == "FP32" else 2), "rho_u_flags": 17, "flags": 1, "F": 12}
        self.face = [local[(a + 1) % 3] * local[(a + 2) % 3] for a in range(3)]
        cap = max([self.face[a] for a in range(3) if D[a] > 1] or [0]) * max(self.per_cell["fi"], 17)
        self.np_buffers = {k: np.zeros(cap, dtype=np.uint8) for k in ("send_p", "send_m", "recv_p", "recv_m")}
        self.buffers = {k: torch.from_numpy(v) for k, v in self.np_buffers.items()} if D[0] * D[1] * D[2] > 1 else None
        self.cfg = self.engine.cfg

    def upload(self, rho, u, flags, F):
        self.rho[...] = rho.ravel(); self.u[...] = u.ravel(); self.flags[...] = flags.ravel()
        if self.F is not None:
            self.F[...] = F.ravel()

    def upload_plane(self, name, z_local, a):
        Lx, Ly, Lz = self.local
        field = getattr(self, name)
        comps = 3 if name in ("u", "F") else 1
        field.reshape(comps, Lz, Ly, Lx)[:, z_local] = np.asarray(a).reshape(comps, Ly, Lx)

    def download(self, names):
        return {k: getattr(self, k) for k in names}

    def initialize(self):
        self.engine.initialize(self.fi, self.rho, self.u, self.flags)

    def collide_planes(self, t, f, z0, z1):
        # the oracle works on whole lattices: restrict it by masking the other planes as solid for this call
        Lx, Ly, Lz = self.local
        hz = 1 if self.cfg.Dz > 1 else 0
        z0, z1 = max(z0, hz), min(z1, Lz - hz)
        if z0 >= z1:
            return
        saved = self.flags.copy()
        fl = self.flags.reshape(Lz, Ly, Lx)
        fl[:z0] = 1; fl[z1:] = 1
        # neighbour flags matter to moving boundaries only; those cases run the un-split path in the tests
        self.engine.stream_collide(self.fi, self.rho, self.u, self.flags, t, f[0], f[1], f[2], self.F)
        self.flags[...] = saved

    def update_fields(self, t, f):
        self.engine.update_fields(self.fi, self.rho, self.u, self.flags, t, f[0], f[1], f[2], self.F)

    def update_force_field(self, t):
        self.engine.update_force_field(self.fi, self.flags, t, self.F)

    def _transfer(self, what, t, insert, axis):
        p = self.np_buffers["recv_p" if insert else "send_p"]
        m = self.np_buffers["recv_m" if insert else "send_m"]
        if what == "fi": self.engine.transfer_fi(insert, axis, t, p, m, self.fi)
        elif what == "rho_u_flags": self.engine.transfer_rho_u_flags(insert, axis, t, p, m, self.rho, self.u, self.flags)
        elif what == "F": self.engine.transfer_F(insert, axis, t, p, m, self.F)

    def pack(self, what, t, axis=2): self._transfer(what, t, 0, axis)
    def unpack(self, what, t, axis=2): self._transfer(what, t, 1, axis)
    def halo_bytes(self, what, axis=2): return self.face[axis] * self.per_cell[what]
