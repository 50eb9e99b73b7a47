"""GPU: the C++ host (cfd_amd/host, the reference's LBM class surface over the C ABI) runs the reference's setups and
agrees with the Python mirror; its VTK export has the layout visualizer.py reads."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import cfd_amd
from cfd_amd import setups

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfd_amd", "host", "bin")


def run_example(name, timeout=300, env=None):
    exe = os.path.join(BIN, name)
    assert os.path.exists(exe), exe + " is missing: make -C cfd_amd/host"
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout
    return r.stdout


def test_taylor_green_example_matches_python_mirror(gpu):
    # 100 steps: the flow is still laminar, so last-ulp differences of the initial field (libm cosf vs numpy) stay small
    out = run_example("taylor_green", env={"LBM_EXAMPLE_STEPS": "100"})
    value = float([l for l in out.splitlines() if re.fullmatch(r"-?[0-9.eE+-]+", l.strip())][0])
    with cfd_amd.LBM((128, 128, 128), 0.01) as lbm:
        setups.taylor_green(lbm)
        lbm.run(100)
        lbm.read_from_device()
        want = float(lbm.u[0, 64, 64, 64])
        full = lbm.u.copy()
    assert abs(value - want) <= 1e-4 * max(abs(want), 1e-3), (value, want)
    # the exported VTK: ASCII header, then big-endian float triples in x-fastest order
    path = os.path.join(BIN, "export", "u-000000100.vtk")
    raw = open(path, "rb").read()
    head_end = raw.index(b"LOOKUP_TABLE default\n") + len(b"LOOKUP_TABLE default\n")
    head = raw[:head_end].decode()
    assert "BINARY" in head and "DATASET STRUCTURED_POINTS" in head and "DIMENSIONS 128 128 128" in head and "SCALARS data float 3" in head
    data = np.frombuffer(raw[head_end:], dtype=">f4").reshape(128, 128, 128, 3)
    assert np.abs(data[64, 64, 64, 0] - want) <= 1e-4 * max(abs(want), 1e-3)
    err = np.linalg.norm((np.moveaxis(data, 3, 0) - full).ravel().astype(np.float64)) / np.linalg.norm(full.ravel().astype(np.float64))
    assert err < 1e-4
    status = open(os.path.join(BIN, "export", "status-000000100.txt")).read()
    assert "Grid Resolution = 128 x 128 x 128 = 2097152\n" in status and "LBM Type = D3Q19 SRT (FP32/FP32)\n" in status
    assert re.search(r"Average MLUPs/s = \d+\n", status) and "Kinematic Viscosity = 1.00000000E-2\n" in status and "Volume Force" not in status


def test_slabs_in_one_process_take_two_steps_at_a_time(gpu):
    """LBM::run on z-slabs (two domains on the one GPU, lbmhip_halo_exchange_local): with the inner planes of two steps in one launch per
    domain (LBMHIP_FUSE=1 forces that on this small lattice) the velocity field has the bits of the undivided lattice stepped one launch at a time."""
    checksum = lambda out: [l for l in out.splitlines() if l.startswith("u checksum ")][0]
    whole = checksum(run_example("taylor_green", env={"LBM_EXAMPLE_STEPS": "51", "LBMHIP_FUSE": "0"}))
    slabs = checksum(run_example("taylor_green", env={"LBM_EXAMPLE_STEPS": "51", "LBMHIP_FUSE": "1", "LBM_EXAMPLE_DOMAINS": "1x1x2"}))
    paired_whole = checksum(run_example("taylor_green", env={"LBM_EXAMPLE_STEPS": "51", "LBMHIP_FUSE": "1"}))
    assert whole == slabs == paired_whole, (whole, slabs, paired_whole)


def test_cavity_example_runs(gpu):
    out = run_example("cavity")
    value = float([l for l in out.splitlines() if re.fullmatch(r"-?[0-9.eE+-]+", l.strip())][0])
    with cfd_amd.LBM((128, 128, 128), setups.nu_from_Re(1000.0, 126.0, 0.1), features=("MOVING_BOUNDARIES",)) as lbm:
        setups.lid_cavity(lbm)
        lbm.run(1000)
        lbm.read_from_device()
        want = float(lbm.u[1, 126, 64, 64])
    assert value == pytest.approx(want, rel=1e-6, abs=1e-9)  # identical inputs: identical result


def test_gui_template_example(gpu, tmp_path):
    """The setup main.py generates: resolution(), voxelize_stl, TYPE_E faces, key_9 export, run loop, write_status."""
    from cfd_amd import mesh as M
    stl = str(tmp_path / "body.stl")
    M.write_stl(stl, *M.torus())
    out = run_example("gui_template", env={"LBM_EXAMPLE_STL": stl, "LBM_EXAMPLE_STEPS": "60"})
    line = [l for l in out.splitlines() if l.startswith("grid ")][0]
    m = re.fullmatch(r"grid (\d+)x(\d+)x(\d+) solid (\d+) t (\d+)", line)
    nx, ny, nz, solid, t = (int(v) for v in m.groups())
    assert nx == ny == nz and 70 <= nx <= 80       # 24 MB at 55 B/cell (D3Q19 FP16S)
    assert solid > 500 and t == 60
    for name, kind, comps in (("u", "float", 3), ("rho", "float", 1), ("flags", "unsigned_char", 1)):
        raw = open(os.path.join(BIN, "export", "%s-000000000.vtk" % name), "rb").read()
        head = raw[:raw.index(b"LOOKUP_TABLE default\n")].decode()
        assert "SCALARS data %s %d" % (kind, comps) in head and "DIMENSIONS %d %d %d" % (nx, ny, nz) in head
    assert os.path.exists(os.path.join(BIN, "export", "status-000000060.txt"))
    raw = open(os.path.join(BIN, "export", "mesh-000000000.vtk"), "rb").read()   # POLYDATA: 1024 triangles of the torus
    assert raw.startswith(b"# vtk DataFile Version 3.0\nFluidX3D mesh-000000000.vtk\nBINARY\nDATASET POLYDATA\nPOINTS 3072 float\n")
    k = raw.index(b"POLYGONS 1024 4096\n")
    pts = np.frombuffer(raw[raw.index(b"float\n") + 6:k], dtype=">f4").reshape(-1, 3)
    assert pts.shape == (3072, 3) and np.abs(pts).max() < 0.5 * nx and abs(float(np.ptp(pts, axis=0).max()) - 0.5 * nz) < 0.05
    tri = np.frombuffer(raw[k + len(b"POLYGONS 1024 4096\n"):], dtype=">i4").reshape(-1, 4)
    assert (tri[:, 0] == 3).all() and np.array_equal(tri[:, 1:].ravel(), np.arange(3072))


def test_fluid_wrapper_as_solver_py_binds_it(gpu, tmp_path):
    import ctypes
    from cfd_amd import mesh as M
    lib = ctypes.CDLL(os.path.join(os.path.dirname(BIN.rstrip("/")), "..", "libfluid_wrapper.so"))
    lib.fluid_init.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_char_p]   # solver.py:95-98
    lib.fluid_step.argtypes = [ctypes.c_int]
    ptr = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    lib.fluid_get_velocity.argtypes = [ptr, ptr, ptr]
    stl = str(tmp_path / "pillar.stl")
    M.write_stl(stl, *M.box(0.3, 0.3, 1.0))
    res = 48
    lib.fluid_init(res, 0.01, 0.0005, stl.encode())
    lib.fluid_step(20)
    lib.fluid_step(20)
    vx, vy, vz = (np.zeros(res ** 3, dtype=np.float32) for _ in range(3))
    lib.fluid_get_velocity(vx, vy, vz)
    lib.fluid_cleanup()
    with cfd_amd.LBM((res, res, res), 0.01, features=("VOLUME_FORCE", "UPDATE_FIELDS"), f=(0.0, 0.0, 0.0005)) as lbm:
        lbm.voxelize_stl(stl, size=0.9 * res)
        lbm.run(40)
        lbm.read_from_device()
        assert (lbm.flags == 1).sum() > 1000
        assert np.array_equal(vx.reshape(res, res, res).view(np.uint32), lbm.u[0].view(np.uint32))
        assert np.array_equal(vz.reshape(res, res, res).view(np.uint32), lbm.u[2].view(np.uint32))
    assert vz.max() > 1e-3 and np.isfinite(vx).all()


def test_stokes_drag_example_object_sums(gpu):
    """object_force / object_torque / object_center_of_mass through the C++ host: drag on a staircase sphere in creeping
    flow is within 12 % of Stokes' law (R = 16 cells), points along the flow, and has no torque."""
    out = run_example("stokes_drag")
    w = [l for l in out.splitlines() if l.startswith("force ")][0].split()
    force, stokes, torque, com = [float(v) for v in w[1:4]], float(w[5]), [float(v) for v in w[7:10]], [float(v) for v in w[11:14]]
    assert force[0] < 0 and abs(abs(force[0]) - stokes) / stokes < 0.12
    assert abs(force[1]) < 1e-5 * stokes and abs(force[2]) < 1e-5 * stokes
    assert max(abs(t) for t in torque) < 1e-4 * stokes * 16
    assert max(abs(c) for c in com) < 1e-3


@pytest.mark.parametrize("name,floor", [("benchmark_fp32", 25000), ("benchmark_fp16s", 45000)])
def test_benchmark_example_reports_peak_mlups(gpu, name, floor):
    """The reference's benchmark setup through the C++ host (512^3 at rest, batches of 10 steps).  The floors are far below
    what the kernels reach (35 000 / 68 000 MLUPs/s); they only catch a host that serialises the launches."""
    out = run_example(name, env={"LBM_BENCH_BATCHES": "100"})
    mlups = int([l for l in out.splitlines() if l.startswith("peak_mlups ")][0].split()[1])
    assert mlups > floor, out


def test_python_vtk_writer_matches_the_cpp_host_layout(gpu, tmp_path):
    """The ctypes mirror writes the same legacy-VTK layout as the C++ host (header lines, big-endian interleaved payload)."""
    with cfd_amd.LBM((24, 16, 8), 0.02, features=("FORCE_FIELD",)) as lbm:
        setups.taylor_green(lbm)
        lbm.flags[2, 3, 4] = 1
        lbm.run(7)
        paths = [lbm.write_device_to_vtk(n, str(tmp_path) + "/") for n in ("u", "rho", "flags", "F")]
        assert os.path.basename(paths[0]) == "u-000000007.vtk"
        for p, name, comps, dt in zip(paths, ("u", "rho", "flags", "F"), (3, 1, 1, 3), (">f4", ">f4", np.uint8, ">f4")):
            raw = open(p, "rb").read()
            end = raw.index(b"LOOKUP_TABLE default\n") + len(b"LOOKUP_TABLE default\n")
            head = raw[:end].decode().splitlines()
            assert head[0] == "# vtk DataFile Version 3.0" and head[2] == "BINARY" and head[3] == "DATASET STRUCTURED_POINTS"
            assert head[4] == "DIMENSIONS 24 16 8" and head[5] == "ORIGIN -1.14999998E1 -7.50000000 -3.50000000" and head[6] == "SPACING 1.00000000 1.00000000 1.00000000"
            assert head[7] == "POINT_DATA 3072" and head[8].startswith("SCALARS data ") and head[8].endswith(" %d" % comps)
            data = np.frombuffer(raw[end:], dtype=dt)
            want = getattr(lbm, name)
            want = np.moveaxis(want, 0, -1) if comps == 3 else want
            assert np.array_equal(data.reshape(want.shape), want)


def test_keys_reach_a_running_setup_through_stdin_and_signals(gpu, tmp_path):
    """The GUI drives the reference through its window (run/pause 'P', export '9', quit); here the same keys arrive on
    standard input or as signals and show up in the setup's loop through running / key_P / key_9."""
    import glob
    import signal
    import time
    from cfd_amd import mesh as M
    stl = str(tmp_path / "body.stl")
    M.write_stl(stl, *M.torus())
    exe = os.path.join(BIN, "gui_template")
    export = os.path.join(BIN, "export")
    for f in glob.glob(os.path.join(export, "u-*.vtk")):
        os.remove(f)
    p = subprocess.Popen([exe], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         env=dict(os.environ, LBM_EXAMPLE_STL=stl, LBM_EXAMPLE_STEPS="100000000"))
    try:
        def exports():
            return sorted(glob.glob(os.path.join(export, "u-*.vtk")))
        deadline = time.time() + 60
        while not exports() and time.time() < deadline:   # the example exports once at t = 0 by itself
            time.sleep(0.05)
        assert len(exports()) == 1
        p.stdin.write(b"9"); p.stdin.flush()               # export trigger through stdin
        while len(exports()) < 2 and time.time() < deadline:
            time.sleep(0.05)
        assert len(exports()) == 2
        p.send_signal(signal.SIGUSR2)                      # pause
        time.sleep(0.3)
        p.send_signal(signal.SIGUSR1)                      # export while paused: the loop still serves key_9
        while len(exports()) < 3 and time.time() < deadline:
            time.sleep(0.05)
        assert len(exports()) == 3
        t_paused = int(re.search(r"u-(\d+)\.vtk", exports()[-1]).group(1))
        time.sleep(0.3)
        stamp = os.stat(exports()[-1]).st_mtime_ns
        p.stdin.write(b"9"); p.stdin.flush()               # a second export while paused lands on the same time step: same file, rewritten
        while os.stat(exports()[-1]).st_mtime_ns == stamp and time.time() < deadline:
            time.sleep(0.05)
        assert len(exports()) == 3 and os.stat(exports()[-1]).st_mtime_ns != stamp
        assert int(re.search(r"u-(\d+)\.vtk", exports()[-1]).group(1)) == t_paused     # no step was taken while paused
        p.stdin.write(b"PQ"); p.stdin.flush()              # resume, then quit
        out, _ = p.communicate(timeout=60)
        assert p.returncode == 0 and "grid " in out.decode()
    finally:
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("case,message", [
    (1, "Viscosity cannot be 0"),
    (2, "Viscosity cannot be negative"),
    (3, "VOLUME_FORCE is not enabled"),
    (4, "EQUILIBRIUM_BOUNDARIES is not enabled"),
    (5, "Grid point number is 0"),
    (6, "does not exist"),
    (7, "0 LBM grid domains"),
])
def test_misconfigured_setups_end_like_the_reference(gpu, case, message):
    """print_error of the reference: one '| Error:' line and exit status 1 (FluidX3D-master/src/utilities.hpp:4159-4166);
    the messages are those of lbm.cpp:1038-1124 and of the sanity checks before initialize."""
    exe = os.path.join(BIN, "error_cases")
    assert os.path.exists(exe), exe + " is missing: make -C cfd_amd/host"
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=dict(os.environ, LBM_ERROR_CASE=str(case)))
    assert r.returncode == 1, r.stdout
    assert "| Error: " in r.stdout and message in r.stdout, r.stdout
    good = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=dict(os.environ, LBM_ERROR_CASE="0")) if case == 1 else None
    if good is not None:
        assert good.returncode == 0 and "ran 1 step" in good.stdout, good.stdout


@pytest.mark.parametrize("storage", ["FP32", "FP16S"])
def test_cpp_slab_runner_over_rccl(gpu, storage):
    """cfd_amd/host/slab_rccl.cpp: one process per GPU in C++ (C ABI + RCCL, no Python in the loop).  With one rank that is
    its own z-neighbour the overlapped exchange must reproduce the periodic lattice bit for bit."""
    out = run_example("slab_rccl", timeout=200, env=dict(LBM_SLAB_SELF="1", LBM_SLAB_N="128", LBM_SLAB_STEPS="60", LBM_SLAB_STORAGE=storage))
    assert "halo exchange over RCCL, overlapped" in out and "bit-identical" in out, out
    # the same with the steps taken two at a time (automatic only from 512 x 512 planes on; forced here), an odd number of them
    out = run_example("slab_rccl", timeout=200, env=dict(LBM_SLAB_SELF="1", LBM_SLAB_N="128", LBM_SLAB_STEPS="61", LBM_SLAB_STORAGE=storage, LBMHIP_FUSE="1"))
    assert "steps two at a time" in out and "bit-identical" in out, out


def test_export_files_equal_the_reference_format_fixture(gpu, tmp_path):
    """The C++ host's VTK and status files against files worked out from the reference's format definition alone
    (tests/golden/make_vtk_golden.py; FluidX3D-master/src/lbm.hpp:298-340, lbm.cpp:104-118, 1599-1643): byte for byte,
    except the clock-dependent status lines, which must match the fixture's patterns."""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vtk_fixture")
    out = str(tmp_path) + "/"
    for pinned in ("1", "0"):  # host copies of the fields in page-locked or in ordinary memory: the same files
        text = run_example("vtk_fixture", env={"LBM_EXAMPLE_OUT": out, "LBM_PINNED": pinned})
        assert "pinned " + pinned in text
        for name in ("rho", "u", "flags", "F"):
            file = "%s-000000000.vtk" % name
            assert open(os.path.join(out, file), "rb").read() == open(os.path.join(golden, file), "rb").read(), file
    got = open(os.path.join(out, "status-000000000.txt")).read().splitlines()
    want = open(os.path.join(golden, "status-000000000.txt")).read().splitlines()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        pattern = "".join(part if k % 2 else re.escape(part) for k, part in enumerate(re.split(r"<<<|>>>", w)))
        assert re.fullmatch(pattern, g), (g, w)
    # the Python mirror writes the same bytes
    with cfd_amd.LBM((4, 3, 2), 0.125, features=("VOLUME_FORCE", "FORCE_FIELD"), f=(0.0, 0.0, 0.0009765625)) as lbm:
        n = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
        lbm.rho[...] = 1.0 + n / 64.0
        lbm.u[0], lbm.u[1], lbm.u[2] = n / 128.0, -n / 256.0, n / 512.0
        lbm.flags[...] = ((np.arange(24) * 7) % 256).astype(np.uint8).reshape(2, 3, 4)
        lbm.F[0], lbm.F[1], lbm.F[2] = n / 16.0, -n / 32.0, n / 8.0
        if hasattr(lbm, "write_host_to_vtk"):
            for name in ("rho", "u", "flags", "F"):
                p = lbm.write_host_to_vtk(name, out + "py/", units=(2.0, 64.0, 0.5))
                assert open(p, "rb").read() == open(os.path.join(golden, os.path.basename(p)), "rb").read(), name


@pytest.mark.parametrize("sig", ["SIGTERM", "SIGINT"])
def test_stop_signal_ends_a_setup_that_ignores_running(gpu, sig):
    """`while(lbm.get_t()<T) lbm.run(n);` without a look at `running` (legal in the reference, which does not intercept
    these signals): the first SIGTERM / SIGINT stops the time stepping, and the process ends by itself after the grace
    period with status 128+signal instead of spinning until it is killed."""
    import signal
    import time
    p = subprocess.Popen([os.path.join(BIN, "stubborn")], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         env=dict(os.environ, LBM_STOP_GRACE_MS="300"))
    try:
        deadline = time.time() + 60
        line = b""
        while b"started" not in line and time.time() < deadline:
            line = p.stdout.readline()
        assert b"started" in line
        time.sleep(0.5)
        assert p.poll() is None
        signum = getattr(signal, sig)
        p.send_signal(signum)
        assert p.wait(timeout=10) == 128 + signum
    finally:
        if p.poll() is None:
            p.kill()
