"""GPU: the reference's own validation setups with their stated outcomes, and an analytic decay - checks of the physics
that do not involve the CPU oracle at all (FluidX3D-master/src/setup.cpp.bak:84-144 Poiseuille flow, "typical expected
L2 errors: 2-5%"; :148-187 Stokes drag lives in tests/test_host_cpp_gpu.py)."""
import numpy as np
import pytest

import cfd_amd

pytestmark = pytest.mark.gpu


def _l2(num, ref):
    return float(np.sqrt(((num - ref) ** 2).sum() / (ref ** 2).sum()))


def test_poiseuille_2d_channel(gpu, storage="FP32"):
    """setup.cpp.bak:84-144 with D2Q9: plates at y = 0 and Ny-1, tau = 1, force 2 u rho nu / R^2; parabola u_max (R^2 - r^2) / R^2.
    FP32 only: with 16-bit DDFs the per-step forcing increment of this setup (6e-6) is of the size of one unit in the last
    place of the stored values and the profile comes out 20 % low - a property of the storage format, which the HIP path
    shares bit for bit with the reference's kernels."""
    R, umax, tau = 31, 0.1, 1.0
    nu = (tau - 0.5) / 3.0
    H = 2 * (R + 1)
    f = 2.0 * umax * 1.0 * nu / R ** 2
    with cfd_amd.LBM((8, H, 1), nu, Q=9, storage=storage, features=("VOLUME_FORCE",), f=(f, 0.0, 0.0)) as lbm:
        lbm.flags[0, 0, :] = 1; lbm.flags[0, -1, :] = 1
        lbm.run(30000)
        lbm.read_from_device()
        y = np.arange(1, H - 1)
        r = y + 0.5 - 0.5 * H
        uref = umax * (R ** 2 - r ** 2) / R ** 2
        unum = np.sqrt(lbm.u[0, 0, 1:-1, 4].astype(np.float64) ** 2 + lbm.u[1, 0, 1:-1, 4].astype(np.float64) ** 2)
        err = _l2(unum, uref)
        assert err < 0.05, err          # the reference quotes 2-5 %
        assert abs(unum.max() / umax - 1.0) < 0.03


def test_poiseuille_3d_pipe_trt(gpu):
    """setup.cpp.bak:84-144 in 3D: a pipe along y, force 4 u rho nu / R^2, checked on the mid-plane inside radius R."""
    R, umax, tau = 15, 0.1, 1.0
    nu = (tau - 0.5) / 3.0
    H = 2 * (R + 1)
    f = 4.0 * umax * 1.0 * nu / R ** 2
    with cfd_amd.LBM((H, 4, H), nu, Q=19, collision="TRT", features=("VOLUME_FORCE",), f=(0.0, f, 0.0)) as lbm:
        z, x = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
        rr = np.sqrt((x + 0.5 - 0.5 * H) ** 2 + (z + 0.5 - 0.5 * H) ** 2)
        solid = rr >= 0.5 * H - 1.0          # cylinder(...) of radius 0.5*min(Nx,Nz)-1 is fluid
        lbm.flags[:, :, :] = solid[:, None, :].astype(np.uint8)
        lbm.run(8000)
        lbm.read_from_device()
        inside = rr < R
        unum = np.sqrt((lbm.u[:, :, 2, :].astype(np.float64) ** 2).sum(axis=0))[inside]
        uref = (umax * (R ** 2 - rr ** 2) / R ** 2)[inside]
        assert _l2(unum, uref) < 0.05


@pytest.mark.parametrize("Q,storage", [(9, "FP32"), (19, "FP32"), (19, "FP16C")])
def test_taylor_green_decay_rate(gpu, Q, storage):
    """A 2-D Taylor-Green vortex decays as exp(-2 nu k^2 t) (k = 2 pi / N): the kinematic viscosity the collision
    operator realises is the one asked for."""
    N, nu, u0, T = 64, 0.02, 0.04, 600
    shape = (N, N, 1) if Q == 9 else (N, N, 4)
    with cfd_amd.LBM(shape, nu, Q=Q, storage=storage) as lbm:
        k = 2.0 * np.pi / N
        y, x = np.meshgrid(np.arange(N) + 0.5, np.arange(N) + 0.5, indexing="ij")
        ux = (u0 * np.cos(k * x) * np.sin(k * y)).astype(np.float32)
        uy = (-u0 * np.sin(k * x) * np.cos(k * y)).astype(np.float32)
        lbm.u[0] = ux[None]; lbm.u[1] = uy[None]
        lbm.rho[...] = (1.0 - 0.75 * u0 ** 2 * (np.cos(2 * k * x) + np.cos(2 * k * y)))[None].astype(np.float32)
        lbm.run(T)
        lbm.read_from_device()
        amp = float((lbm.u[0, 0].astype(np.float64) * ux).sum() / (ux.astype(np.float64) ** 2).sum())   # projection on the initial mode
        want = np.exp(-2.0 * nu * k * k * T)
        assert abs(amp / want - 1.0) < 0.01, (amp, want)
