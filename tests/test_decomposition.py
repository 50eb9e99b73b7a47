"""CPU: host-side decomposition arithmetic against a brute-force statement of the reference's rules."""
import numpy as np
import pytest

from cfd_amd import decomposition as dc


@pytest.mark.parametrize("N,D", [((32, 32, 64), (1, 1, 2)), ((33, 30, 17), (2, 3, 1)), ((24, 24, 24), (2, 2, 2)), ((16, 8, 40), (1, 1, 5))])
def test_slices_cover_the_lattice_once(N, D):
    Nr = dc.rounded_size(N, D)
    assert all(Nr[k] % D[k] == 0 and Nr[k] <= N[k] for k in range(3))
    seen = np.zeros((Nr[2], Nr[1], Nr[0]), dtype=np.int32)
    L = dc.local_size(N, D)
    for d in range(D[0] * D[1] * D[2]):
        g, l = dc.slices(d, N, D)
        seen[g] += 1
        off = dc.domain_offset(d, N, D)
        for k, ax in enumerate((2, 1, 0)):  # the local slice starts after the halo layer and maps to the global offset
            assert g[k].start == off[ax] + l[k].start
            assert l[k].stop <= L[ax]
    assert (seen == 1).all()


@pytest.mark.parametrize("N,D", [((12, 8, 10), (2, 1, 2)), ((9, 9, 9), (3, 3, 3))])
def test_locate_agrees_with_slices(N, D):
    Nr = dc.rounded_size(N, D)
    L = dc.local_size(N, D)
    owner = np.full(Nr[0] * Nr[1] * Nr[2], -1)
    local = np.full(Nr[0] * Nr[1] * Nr[2], -1)
    gidx = np.arange(Nr[0] * Nr[1] * Nr[2]).reshape(Nr[2], Nr[1], Nr[0])
    lidx = np.arange(L[0] * L[1] * L[2]).reshape(L[2], L[1], L[0])
    for d in range(D[0] * D[1] * D[2]):
        g, l = dc.slices(d, N, D)
        owner[gidx[g].ravel()] = d
        local[gidx[g].ravel()] = lidx[l].ravel()
    for n in range(0, owner.size, 7):
        assert dc.locate(n, N, D) == (owner[n], local[n])


def test_neighbours_wrap_periodically():
    D = (2, 1, 4)
    assert dc.neighbour(dc.domain_index((1, 0, 3), D), D, 2) == dc.domain_index((1, 0, 0), D)
    assert dc.neighbour(dc.domain_index((1, 0, 0), D), D, 2, -1) == dc.domain_index((1, 0, 3), D)
    assert dc.neighbour(dc.domain_index((1, 0, 2), D), D, 0) == dc.domain_index((0, 0, 2), D)
    assert dc.halo_bytes_per_cell("fi", 19, 4) == 20 and dc.halo_bytes_per_cell("fi", 27, 2) == 18 and dc.halo_bytes_per_cell("F", 19, 4) == 12
