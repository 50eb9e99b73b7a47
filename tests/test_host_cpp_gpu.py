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
    assert "Grid Resolution = (128, 128, 128)" in status and "Average MLUPs/s" in status


def test_cavity_example_runs(gpu):
    out = run_example("cavity")
    value = float([l for l in out.splitlines() if re.fullmatch(r"-?[0-9.eE+-]+", l.strip())][0])
    with cfd_amd.LBM((128, 128, 128), setups.nu_from_Re(1000.0, 126.0, 0.1), features=("MOVING_BOUNDARIES",)) as lbm:
        setups.lid_cavity(lbm)
        lbm.run(1000)
        lbm.read_from_device()
        want = float(lbm.u[1, 126, 64, 64])
    assert value == pytest.approx(want, rel=1e-6, abs=1e-9)  # identical inputs: identical result
