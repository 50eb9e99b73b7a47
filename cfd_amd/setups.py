"""Synthetic setups (host side): deterministic initial fields for the benchmark and parity configurations.

Every function fills the global fields of an LBM-like object exposing N=(Nx,Ny,Nz) and numpy arrays
rho[z,y,x], u[3,z,y,x], flags[z,y,x] (and F[3,z,y,x] with FORCE_FIELD).  The formulae follow the setups
the reference ships in FluidX3D-master/src/setup.cpp.bak.
"""
import numpy as np

TYPE_S, TYPE_E, TYPE_X, TYPE_Y = 0x01, 0x02, 0x40, 0x80


def _axes(lbm):
    """Global x, y, z coordinates of the cells held by `lbm` (its fields may cover only a slab of the lattice)."""
    nz, ny, nx = lbm.rho.shape
    ox, oy, oz = getattr(lbm, "origin", (0, 0, 0))
    return np.arange(ox, ox + nx), np.arange(oy, oy + ny), np.arange(oz, oz + nz)


def _grid(lbm):
    xa, ya, za = _axes(lbm)
    z, y, x = np.meshgrid(za, ya, xa, indexing="ij")
    return x, y, z


def nu_from_Re(Re, x, u):
    """FluidX3D-master/src/units.hpp nu_from_Re: nu = x*u/Re, evaluated in float."""
    return float(np.float32(np.float32(x) * np.float32(u)) / np.float32(Re))


def taylor_green(lbm, A=0.25, periodicity=1):
    """3D Taylor-Green vortices, fully periodic (setup.cpp.bak:40-59).  Every factor depends on one coordinate only,
    so the field is built from 1-D float32 factors multiplied in the order the reference writes them."""
    Nx, Ny, Nz = lbm.N
    xa, ya, za = _axes(lbm)
    f32 = np.float32
    a, b, c = f32(Nx) / f32(periodicity), f32(Ny) / f32(periodicity), f32(Nz) / f32(periodicity)
    fx = xa.astype(f32) + f32(0.5) - f32(0.5) * f32(Nx)
    fy = ya.astype(f32) + f32(0.5) - f32(0.5) * f32(Ny)
    fz = za.astype(f32) + f32(0.5) - f32(0.5) * f32(Nz)
    pif = f32(3.1415927)
    A = f32(A)
    ax, ay, az = f32(2.0) * pif * fx / a, f32(2.0) * pif * fy / b, f32(2.0) * pif * fz / c
    cx, sx = np.cos(ax)[None, None, :], np.sin(ax)[None, None, :]
    cy, sy = np.cos(ay)[None, :, None], np.sin(ay)[None, :, None]
    cz, sz = np.cos(az)[:, None, None], np.sin(az)[:, None, None]
    lbm.u[0] = ((A * cx) * sy) * sz
    lbm.u[1] = ((-A * sx) * cy) * sz
    lbm.u[2] = ((A * sx) * sy) * cz
    c4x = np.cos(f32(4.0) * pif * fx / a)[None, None, :]
    c4y = np.cos(f32(4.0) * pif * fy / b)[None, :, None]
    lbm.rho[...] = f32(1.0) - A * A * f32(3.0) / f32(4.0) * (c4x + c4y) + f32(0.0) * sz
    lbm.flags[...] = 0


def lid_cavity(lbm, u_lid=0.1):
    """Lid-driven cavity: all six faces solid, the lid z=Nz-1 moves in +y (setup.cpp.bak:236-249).
    Needs MOVING_BOUNDARIES."""
    Nx, Ny, Nz = lbm.N
    x, y, z = _grid(lbm)
    lbm.rho[...] = 1.0
    lbm.u[...] = 0.0
    lbm.u[1][z == Nz - 1] = np.float32(u_lid)
    wall = (x == 0) | (x == Nx - 1) | (y == 0) | (y == Ny - 1) | (z == 0) | (z == Nz - 1)
    lbm.flags[...] = np.where(wall, TYPE_S, 0).astype(np.uint8)


def obstacle_mask(Nx, Ny, seed=42, n_pillars=12):
    """Seeded 2-D obstacle mask (pillars of random radius) used for the extruded channel geometry."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(Ny), np.arange(Nx), indexing="ij")
    mask = np.zeros((Ny, Nx), dtype=bool)
    for _ in range(n_pillars):
        cx, cy = rng.uniform(0.1, 0.9) * Nx, rng.uniform(0.2, 0.8) * Ny
        r = rng.uniform(0.02, 0.06) * min(Nx, Ny)
        mask |= (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
    return mask


def channel_flag_plane(Nx, Ny, Nz, z, seed=42, mask=None):
    """Flags of lattice plane z of the channel below ([y, x]): pillars and y walls, the first and last plane all solid."""
    if z == 0 or z == Nz - 1:
        return np.full((Ny, Nx), TYPE_S, dtype=np.uint8)
    plane = (obstacle_mask(Nx, Ny, seed) if mask is None else mask).astype(np.uint8) * np.uint8(TYPE_S)
    plane[0] = plane[-1] = TYPE_S
    return plane


def channel(lbm, seed=42, u0=0.0):
    """Force-driven channel, periodic in x, solid walls in y and z, plus an obstacle mask extruded along z.
    Needs VOLUME_FORCE (the force comes from the LBM constructor)."""
    Nx, Ny, Nz = lbm.N
    xa, ya, za = _axes(lbm)
    lbm.rho[...] = 1.0
    lbm.u[...] = 0.0
    lbm.u[0][...] = np.float32(u0)
    mask = obstacle_mask(Nx, Ny, seed)
    for k, z in enumerate(za):
        lbm.flags[k] = channel_flag_plane(Nx, Ny, Nz, int(z), seed, mask)[ya[0]:ya[-1] + 1, xa[0]:xa[-1] + 1]


def inflow_box(lbm, u_in=0.075, radius_frac=0.15):
    """What the GUI's generated setup does (main.py:145-198): u.x preset everywhere outside solids, all six
    faces TYPE_E, a solid body in the middle (a sphere stands in for the voxelized STL).
    Needs EQUILIBRIUM_BOUNDARIES."""
    Nx, Ny, Nz = lbm.N
    x, y, z = _grid(lbm)
    r2 = (x - 0.5 * Nx + 0.5) ** 2 + (y - 0.5 * Ny + 0.5) ** 2 + (z - 0.5 * Nz + 0.5) ** 2
    solid = r2 <= (radius_frac * min(Nx, Ny, Nz)) ** 2
    lbm.rho[...] = 1.0
    lbm.u[...] = 0.0
    lbm.u[0][~solid] = np.float32(u_in)
    flags = np.where(solid, TYPE_S, 0).astype(np.uint8)
    face = (x == 0) | (x == Nx - 1) | (y == 0) | (y == Ny - 1) | (z == 0) | (z == Nz - 1)
    flags[face] = TYPE_E
    lbm.flags[...] = flags


def sphere_force(lbm, u_in=0.05, radius_frac=0.2):
    """Sphere flagged TYPE_S|TYPE_X in a box with TYPE_E faces (Stokes-drag style, setup.cpp.bak:148-187).
    Needs EQUILIBRIUM_BOUNDARIES; FORCE_FIELD gives the boundary force on the sphere."""
    Nx, Ny, Nz = lbm.N
    x, y, z = _grid(lbm)
    r2 = (x - 0.5 * Nx + 0.5) ** 2 + (y - 0.5 * Ny + 0.5) ** 2 + (z - 0.5 * Nz + 0.5) ** 2
    solid = r2 <= (radius_frac * min(Nx, Ny, Nz)) ** 2
    lbm.rho[...] = 1.0
    lbm.u[...] = 0.0
    lbm.u[0][~solid] = np.float32(-u_in)
    flags = np.zeros(x.shape, dtype=np.uint8)
    face = (x == 0) | (x == Nx - 1) | (y == 0) | (y == Ny - 1) | (z == 0) | (z == Nz - 1)
    flags[face] = TYPE_E
    flags[solid] = TYPE_S | TYPE_X
    lbm.flags[...] = flags
