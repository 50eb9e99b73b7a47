"""cfd_amd - MI355X-native lattice-Boltzmann time step (the FluidX3D core of Noelok/CFD).

The product is the C-ABI library cfd_amd/liblbmhip.so (include/lbmhip.h) with hand-written HIP kernels for
gfx950.  This package holds its sources (csrc/), the C++ host class (host/) and a thin Python mirror of the
reference's LBM class used by the tests and the benchmark.  There is no CPU fallback: importing works without
a GPU, every compute call needs the library and a device.
"""
from .lbm import LBM, LbmHipError, lib, lib_path, FEATURES, STORAGE, COLLISION  # noqa: F401
from . import setups  # noqa: F401
from . import mesh  # noqa: F401
