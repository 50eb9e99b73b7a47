"""Domain decomposition arithmetic of the reference's LBM class (pure host logic, no device access).

FluidX3D-master/src/lbm.cpp:1038-1073: the global lattice is rounded down to a multiple of the domain count per
axis; domain d = x+(y+z*Dy)*Dx holds N/D cells per axis plus a 1-cell halo on both ends of every split axis, at
global offset x*Nx/Dx - 1 (halo included).  FluidX3D-master/src/lbm.hpp:265-288 maps a global cell to
(domain, local index).
"""


def rounded_size(N, D):
    return tuple((int(N[k]) // int(D[k])) * int(D[k]) for k in range(3))


def halo(D):
    return tuple(1 if int(d) > 1 else 0 for d in D)


def local_size(N, D):
    N = rounded_size(N, D)
    H = halo(D)
    return tuple(N[k] // D[k] + 2 * H[k] for k in range(3))


def domain_coordinates(d, D):
    Dx, Dy, _ = D
    return (d % (Dx * Dy)) % Dx, (d % (Dx * Dy)) // Dx, d // (Dx * Dy)


def domain_index(c, D):
    return c[0] + (c[1] + c[2] * D[1]) * D[0]


def domain_offset(d, N, D):
    """Ox, Oy, Oz of domain d (signed; -1 on split axes for the first domain)."""
    N = rounded_size(N, D)
    H = halo(D)
    c = domain_coordinates(d, D)
    return tuple(c[k] * N[k] // D[k] - H[k] for k in range(3))


def neighbour(d, D, axis, step=1):
    """Index of the domain `step` positions further along `axis`, with periodic wrap (lbm.cpp:2130-2132)."""
    c = list(domain_coordinates(d, D))
    c[axis] = (c[axis] + step) % D[axis]
    return domain_index(c, D)


def slices(d, N, D):
    """(global, local) index slices in [z, y, x] order that connect domain d with the global fields."""
    N = rounded_size(N, D)
    H = halo(D)
    c = domain_coordinates(d, D)
    n = tuple(N[k] // D[k] for k in range(3))
    glob = tuple(slice(c[k] * n[k], (c[k] + 1) * n[k]) for k in (2, 1, 0))
    loc = tuple(slice(H[k], H[k] + n[k]) for k in (2, 1, 0))
    return glob, loc


def locate(n, N, D):
    """Global linear cell index -> (domain, local linear index), as Memory_Container::reference does."""
    N = rounded_size(N, D)
    H = halo(D)
    L = local_size(N, D)
    plane = N[0] * N[1]
    t = n % plane
    g = (t % N[0], t // N[0], n // plane)
    per = tuple(N[k] // D[k] for k in range(3))
    p = tuple(g[k] % per[k] for k in range(3))
    c = tuple(g[k] // per[k] for k in range(3))
    local = (p[0] + H[0]) + ((p[1] + H[1]) + (p[2] + H[2]) * L[1]) * L[0]
    return domain_index(c, D), local


def halo_bytes_per_cell(what, Q, fpxx_bytes):
    """LBM::communicate_* (lbm.cpp:2184-2194): 'fi', 'rho_u_flags', 'flags' or 'F'."""
    transfers = {9: 3, 15: 5, 19: 5, 27: 9}[Q]
    return {"fi": transfers * fpxx_bytes, "rho_u_flags": 17, "flags": 1, "F": 12}[what]
