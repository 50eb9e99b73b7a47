"""CPU: the C-ABI library loads, exports every symbol include/lbmhip.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

import cfd_amd
from cfd_amd import lbm as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lbmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbmhip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = cfd_amd.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "liblbmhip.so does not export " + n


def test_byte_models_match_the_reference():
    lib = cfd_amd.lib()
    bw = lib.lbmhip_bandwidth_bytes_per_cell
    F = L.FEATURES
    assert bw(19, 0, 0) == 153 and bw(19, 1, 0) == 77 and bw(19, 2, 0) == 77           # README.md:509
    assert bw(19, 0, F["MOVING_BOUNDARIES"]) == 171 and bw(19, 0, F["UPDATE_FIELDS"]) == 169
    assert bw(27, 0, 0) == 217 and bw(27, 2, F["FORCE_FIELD"]) == 121
    cap = lib.lbmhip_bytes_per_cell_device
    assert cap(19, 0, 0) == 93 and cap(19, 1, 0) == 55 and cap(27, 2, F["FORCE_FIELD"]) == 83  # README.md:508


def test_no_gpu_means_loud_failure_not_fallback():
    lib = cfd_amd.lib()
    n = C.c_int(-1)
    status = lib.lbmhip_device_count(C.byref(n))
    if status == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cfd_amd.LbmHipError):
        cfd_amd.LBM((16, 16, 16), 0.1)
    assert lib.lbmhip_last_error()


def test_config_validation_messages():
    lib = cfd_amd.lib()
    h = C.c_void_p()
    bad = L.Config(0, 16, 16, 1, 1, 1, 0, 0, 0, 19, 0, 0, 0, 0.1, 0)
    assert lib.lbmhip_domain_create(C.byref(bad), C.byref(h)) != 0 and b"Grid point number is 0" in lib.lbmhip_last_error()
    bad = L.Config(16, 16, 16, 1, 1, 1, 0, 0, 0, 19, 0, 0, 0, 0.0, 0)
    assert lib.lbmhip_domain_create(C.byref(bad), C.byref(h)) != 0 and b"Viscosity cannot be 0" in lib.lbmhip_last_error()
    bad = L.Config(16, 16, 16, 1, 1, 1, 0, 0, 0, 13, 0, 0, 0, 0.1, 0)
    assert lib.lbmhip_domain_create(C.byref(bad), C.byref(h)) != 0 and b"velocity set" in lib.lbmhip_last_error()
    bad = L.Config(16, 16, 2, 1, 1, 1, 0, 0, 0, 9, 0, 0, 0, 0.1, 0)
    assert lib.lbmhip_domain_create(C.byref(bad), C.byref(h)) != 0 and b"D2Q9" in lib.lbmhip_last_error()
    assert not h.value


def test_fluid_wrapper_exports_the_symbols_solver_py_binds():
    path = os.path.join(ROOT, "cfd_amd", "libfluid_wrapper.so")
    assert os.path.exists(path), "make -C cfd_amd/host"
    lib = C.CDLL(path)
    for n in ("fluid_init", "fluid_step", "fluid_get_velocity", "fluid_cleanup"):
        assert hasattr(lib, n)
    lib.fluid_cleanup()  # harmless without a simulation
