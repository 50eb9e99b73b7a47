"""GPU: error behaviour of the C ABI.  Every misuse returns a non-zero status with a message from lbmhip_last_error()
(the C++ host turns that into the reference's print_error + exit, FluidX3D-master/src/utilities.hpp:4159-4166); nothing is
launched and the domain stays usable afterwards."""
import ctypes as C

import numpy as np
import pytest

import cfd_amd
from cfd_amd import lbm as L

pytestmark = pytest.mark.gpu


def last(lib):
    return lib.lbmhip_last_error().decode()


def test_misuse_is_reported_and_harmless(gpu):
    lib = cfd_amd.lib()
    with cfd_amd.LBM((16, 16, 16), 0.05) as lbm:
        dom = lbm.domains[0]
        h = dom.h
        buf = np.zeros(16, dtype=np.float32)
        n_rho = dom.N * 4
        # transfers outside a field, also with offset + bytes wrapping around 2^64
        assert lib.lbmhip_upload(h, L.FIELD_RHO, buf.ctypes.data_as(C.c_void_p), C.c_uint64(n_rho - 32), C.c_uint64(64)) != 0 and "exceeds field" in last(lib)
        assert lib.lbmhip_download(h, L.FIELD_RHO, buf.ctypes.data_as(C.c_void_p), C.c_uint64(2 ** 64 - 8), C.c_uint64(64)) != 0 and "exceeds field" in last(lib)
        assert lib.lbmhip_upload(h, 7, buf.ctypes.data_as(C.c_void_p), C.c_uint64(0), C.c_uint64(4)) != 0 and "does not exist" in last(lib)
        assert lib.lbmhip_upload(h, L.FIELD_F, buf.ctypes.data_as(C.c_void_p), C.c_uint64(0), C.c_uint64(4)) != 0 and "does not exist" in last(lib)  # no FORCE_FIELD
        # extensions that were not enabled
        assert lib.lbmhip_update_force_field(h, C.c_uint64(0)) != 0 and "FORCE_FIELD" in last(lib)
        assert lib.lbmhip_reset_force_field(h) != 0 and "FORCE_FIELD" in last(lib)
        assert lib.lbmhip_update_moving_boundaries(h) != 0 and "MOVING_BOUNDARIES" in last(lib)
        sums = (C.c_double * 4)()
        assert lib.lbmhip_object_sum(h, 1, 1, None, None, sums) != 0 and "FORCE_FIELD" in last(lib)
        assert lib.lbmhip_object_sum(h, 5, 1, None, None, sums) != 0 and "what must be" in last(lib)
        # a single domain has no halo
        assert lib.lbmhip_halo_pack(h, L.HALO_FI, 2, C.c_uint64(0)) != 0 and "no halo" in last(lib)
        assert lib.lbmhip_halo_pack(h, L.HALO_FI, 3, C.c_uint64(0)) != 0 and "axis" in last(lib)
        nbytes = C.c_uint64()
        assert lib.lbmhip_halo_bytes(h, 9, 2, C.byref(nbytes)) != 0 and "halo content" in last(lib)
        # plane ranges and knobs
        z = (C.c_uint64(0), C.c_float(0), C.c_float(0), C.c_float(0))
        assert lib.lbmhip_stream_collide_planes(h, *z, 4, 17) != 0 and "outside the lattice" in last(lib)
        assert lib.lbmhip_stream_collide_planes(h, *z, 9, 3) != 0 and "outside the lattice" in last(lib)
        assert lib.lbmhip_set_tuning(h, b"cells_per_thread", 3) != 0 and "cells_per_thread" in last(lib)
        assert lib.lbmhip_set_tuning(h, b"cells_together", 4) != 0 and "cells_together" in last(lib)
        assert lib.lbmhip_set_tuning(h, b"block", 100) != 0 and "block" in last(lib)
        assert lib.lbmhip_set_tuning(h, b"no_such_knob", 1) != 0 and "unknown tuning key" in last(lib)
        v = C.c_int()
        assert lib.lbmhip_get_tuning(h, b"no_such_knob", C.byref(v)) != 0
        # communication-stream bracket out of order
        assert lib.lbmhip_comm_end(h) != 0 and "without" in last(lib)
        assert lib.lbmhip_comm_begin(h) == 0 and lib.lbmhip_comm_begin(h) != 0 and "already" in last(lib)
        assert lib.lbmhip_comm_join(h) != 0 and lib.lbmhip_comm_end(h) == 0 and lib.lbmhip_comm_join(h) == 0
        # voxelizer arguments
        tri = np.zeros(9, dtype=np.float32)
        box = np.zeros(6, dtype=np.float32)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        assert lib.lbmhip_voxelize_mesh(h, 3, C.c_uint64(0), 1, p(tri), p(tri), p(tri), 1, p(box), None, None, None) != 0 and "direction" in last(lib)
        assert lib.lbmhip_voxelize_mesh(h, 0, C.c_uint64(0), 1, p(tri), p(tri), p(tri), 0, p(box), None, None, None) != 0 and "triangles" in last(lib)
        # null handles
        assert lib.lbmhip_stream_collide(None, *z) != 0 and lib.lbmhip_sync(None) != 0 and lib.lbmhip_domain_destroy(None) == 0
        # the domain still works and nothing above wrote to it
        from cfd_amd import setups
        setups.taylor_green(lbm)
        lbm.run(3)
        lbm.read_from_device()
        assert np.isfinite(lbm.u).all() and abs(float(lbm.rho.mean()) - 1.0) < 1e-3
    # a lattice that cannot fit in device memory reports the reference's message (lbm.cpp:150-157) and leaves nothing allocated
    free0, total = C.c_uint64(), C.c_uint64()
    assert lib.lbmhip_device_memory(0, C.byref(free0), C.byref(total)) == 0
    cfg = L.Config(4096, 4096, 4096, 1, 1, 1, 0, 0, 0, 27, 0, 0, 0, 0.1, 0)  # 7.4 TB of DDFs
    hh = C.c_void_p()
    assert lib.lbmhip_domain_create(C.byref(cfg), C.byref(hh)) != 0 and "too large" in last(lib) and not hh.value
    free1 = C.c_uint64()
    assert lib.lbmhip_device_memory(0, C.byref(free1), None) == 0
    assert abs(int(free1.value) - int(free0.value)) < (1 << 30)


def test_halo_buffer_bookkeeping_and_mixed_domain_lists(gpu):
    """Caller-owned staging buffers are given back with the library's own capacity, and the one-process exchange refuses
    domain lists it cannot serve (a missing handle, blocks of different size) instead of copying wrong lengths."""
    lib = cfd_amd.lib()
    with cfd_amd.LBM((16, 16, 16), 0.01, D=(1, 1, 2), devices=[0, 0]) as lbm, cfd_amd.LBM((24, 16, 16), 0.01, D=(1, 1, 2), devices=[0, 0]) as other:
        h = lbm.domains[0].h
        p, own = C.c_void_p(), C.c_uint64()
        assert lib.lbmhip_halo_buffer(h, 0, C.byref(p), C.byref(own)) == 0 and own.value == 16 * 16 * 20
        # caller-owned device memory: the (larger) staging buffers of another domain stand in for it
        big, big_cap = [], C.c_uint64()
        for k in range(4):
            q = C.c_void_p()
            assert lib.lbmhip_halo_buffer(other.domains[0].h, k, C.byref(q), C.byref(big_cap)) == 0
            big.append(q.value)
        assert big_cap.value == 24 * 16 * 20 > own.value
        assert lib.lbmhip_halo_set_buffers(h, *[C.c_void_p(b) for b in big], big_cap) == 0
        cap = C.c_uint64()
        assert lib.lbmhip_halo_buffer(h, 0, C.byref(p), C.byref(cap)) == 0 and cap.value == big_cap.value and p.value == big[0]
        assert lib.lbmhip_halo_set_buffers(h, None, None, None, None, C.c_uint64(0)) == 0
        assert lib.lbmhip_halo_buffer(h, 0, C.byref(p), C.byref(cap)) == 0 and cap.value == own.value and p.value != big[0]
        last = lambda: lib.lbmhip_last_error().decode()
        two = (C.c_void_p * 2)(lbm.domains[0].h, None)
        assert lib.lbmhip_halo_exchange_local(two, 2, 0, 2) != 0 and "null" in last()
        mixed = (C.c_void_p * 2)(lbm.domains[0].h, other.domains[1].h)
        assert lib.lbmhip_halo_exchange_local(mixed, 2, 0, 2) != 0 and "differs from domain 0" in last()
        setups_ok = (C.c_void_p * 2)(lbm.domains[0].h, lbm.domains[1].h)
        for dom in lbm.domains:
            assert lib.lbmhip_halo_pack(dom.h, 0, 2, C.c_uint64(0)) == 0
        assert lib.lbmhip_halo_exchange_local(setups_ok, 2, 0, 2) == 0
        lbm.sync()
