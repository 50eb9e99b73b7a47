"""Parity cases shared by the CPU and GPU tests: (name, constructor arguments, setup, steps)."""
import numpy as np

from cfd_amd import setups

ALL = ("UPDATE_FIELDS", "VOLUME_FORCE", "MOVING_BOUNDARIES", "EQUILIBRIUM_BOUNDARIES", "FORCE_FIELD", "SUBGRID")


def _all_features_setup(lbm):
    """TYPE_E faces, a solid sphere, a moving solid plate and a random force field: touches every branch."""
    setups.inflow_box(lbm)
    Nx, Ny, Nz = lbm.N
    z0, y0, x0 = (5 * Nz) // 8, Ny // 3, Nx // 6
    lbm.flags[z0:z0 + 3, y0:y0 + Ny // 4, x0:x0 + Nx // 6] = 1
    lbm.u[1, z0:z0 + 3, y0:y0 + Ny // 4, x0:x0 + Nx // 6] = 0.05
    if lbm.F is not None:
        rng = np.random.RandomState(1)
        lbm.F[...] = (rng.rand(*lbm.F.shape).astype(np.float32) - 0.5) * 1e-4


def _channel_setup(lbm):
    setups.channel(lbm, seed=42)


def _sphere_setup(lbm):
    setups.sphere_force(lbm)


def _tgv2d(lbm):
    setups.taylor_green(lbm, A=0.2, periodicity=2)
    lbm.u[2] = 0.0


# name -> dict(kw=constructor keywords, setup=callable, steps=int)
CASES = {
    "tgv32_fp32_srt": dict(kw=dict(N=(32, 32, 32), nu=0.01), setup=setups.taylor_green, steps=40),
    "tgv_odd_fp32_srt": dict(kw=dict(N=(40, 24, 20), nu=0.02), setup=setups.taylor_green, steps=30),
    "tgv128x8_fp32_trt": dict(kw=dict(N=(128, 8, 8), nu=0.01, collision="TRT"), setup=setups.taylor_green, steps=30),
    "cavity32_fp32_srt": dict(kw=dict(N=(32, 32, 32), nu=0.0126, features=("MOVING_BOUNDARIES",)), setup=setups.lid_cavity, steps=60),
    "channel_fp32_trt_sgs": dict(kw=dict(N=(32, 24, 16), nu=0.005, collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0)), setup=_channel_setup, steps=60),
    "gui_fp16s_srt": dict(kw=dict(N=(32, 32, 32), nu=0.002, storage="FP16S", features=("UPDATE_FIELDS", "VOLUME_FORCE", "EQUILIBRIUM_BOUNDARIES", "SUBGRID")), setup=setups.inflow_box, steps=50),
    "all_fp16s_trt_2dom": dict(kw=dict(N=(32, 32, 64), nu=0.01, D=(1, 1, 2), storage="FP16S", collision="TRT", features=ALL, f=(1e-4, 0.0, 0.0)), setup=_all_features_setup, steps=50),
    "all_fp32_srt_8dom": dict(kw=dict(N=(32, 32, 32), nu=0.01, D=(2, 2, 2), features=ALL, f=(1e-4, 0.0, 0.0)), setup=_all_features_setup, steps=30),
    "q27_fp16c_force": dict(kw=dict(N=(32, 32, 32), nu=0.02, Q=27, storage="FP16C", features=("VOLUME_FORCE", "FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES")), setup=_sphere_setup, steps=40),
    "q27_fp32_trt_4slab": dict(kw=dict(N=(24, 24, 48), nu=0.02, Q=27, D=(1, 1, 4), collision="TRT"), setup=setups.taylor_green, steps=30),
    "q19_fp16c_srt": dict(kw=dict(N=(32, 32, 32), nu=0.01, storage="FP16C"), setup=setups.taylor_green, steps=40),
    "q15_fp32_srt": dict(kw=dict(N=(24, 24, 24), nu=0.02, Q=15, D=(2, 1, 1)), setup=setups.taylor_green, steps=30),
    "q9_fp32_srt": dict(kw=dict(N=(48, 32, 1), nu=0.02, Q=9), setup=_tgv2d, steps=40),
}


# Long runs: the benchmark configurations at sizes the compiled reference kernels finish in about a minute on all cores,
# over the 1000 steps the tolerance is quoted for.  name -> dict(kw, setup, steps, tolerance = relative L2 bound on rho and u)
LONG_CASES = {
    "long_cavity128_fp32": dict(kw=dict(N=(128, 128, 128), nu=0.0126, features=("MOVING_BOUNDARIES",)), setup=setups.lid_cavity, steps=1000, tolerance=1e-5),
    "long_tgv64_fp32": dict(kw=dict(N=(64, 64, 64), nu=0.01), setup=setups.taylor_green, steps=1000, tolerance=1e-5),
    "long_tgv128_fp32": dict(kw=dict(N=(128, 128, 128), nu=0.01), setup=setups.taylor_green, steps=1000, tolerance=1e-5),
    "long_tgv128_fp16s": dict(kw=dict(N=(128, 128, 128), nu=0.01, storage="FP16S"), setup=setups.taylor_green, steps=1000, tolerance=1e-3),
    "long_channel_trt_sgs_2slab": dict(kw=dict(N=(96, 64, 64), nu=0.005, D=(1, 1, 2), collision="TRT", features=("VOLUME_FORCE", "SUBGRID"), f=(2e-4, 0.0, 0.0)),
                                       setup=_channel_setup, steps=1000, tolerance=1e-5),
    "long_q27_fp16c_force_4slab": dict(kw=dict(N=(48, 48, 96), nu=0.02, Q=27, D=(1, 1, 4), storage="FP16C", features=("VOLUME_FORCE", "FORCE_FIELD", "EQUILIBRIUM_BOUNDARIES")),
                                       setup=_sphere_setup, steps=1000, tolerance=1e-3),
}
