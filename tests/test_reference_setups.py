"""The drop-in claim of the C++ host, checked against the reference's own setup collection.

CPU (needs /root/reference, skipped elsewhere): every setup of FluidX3D-master/src/setup.cpp.bak whose extensions this
build provides compiles UNCHANGED against cfd_amd/host (syntax check; the text is read where it lies, nothing is copied).
GPU: three of those setups, built unchanged by oracle/ref/build_setups.py into oracle/_ref/, run to their own stop
criterion and report what the reference's comments promise."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def test_reference_setups_compile_unchanged():
    if not os.path.exists("/root/reference/FluidX3D-master/src/setup.cpp.bak"):
        pytest.skip("reference sources are not present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compile_reference_setups.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    m = re.search(r"compiled (\d+), failed (\d+), skipped \(extensions not provided\) (\d+)", r.stdout)
    assert m, r.stdout
    compiled, failed, skipped = (int(v) for v in m.groups())
    assert failed == 0 and compiled >= 24, r.stdout   # the 13 skipped ones need SURFACE, TEMPERATURE or PARTICLES


def _run(name, timeout=280):
    exe = os.path.join(REF, "setup_" + name)
    if not os.path.exists(exe):
        pytest.skip(exe + " not built (needs /root/reference at build time)")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_reference_poiseuille_setup_runs_unchanged(gpu):
    out = _run("poiseuille")   # FluidX3D-master/src/setup.cpp.bak:84-144, "typical expected L2 errors: 2-5%"
    m = re.search(r"Poiseuille flow error converged after (\d+) steps to ([0-9.]+)%", out)
    assert m, out[-1000:]
    assert float(m.group(2)) < 5.0


@pytest.mark.gpu
def test_reference_stokes_drag_setup_runs_unchanged(gpu):
    out = _run("stokes_drag")  # setup.cpp.bak:148-187: object_force on a sphere against Stokes' law
    m = re.search(r"Error converged after (\d+) steps to ([0-9.]+)%", out)
    assert m, out[-1000:]
    assert float(m.group(2)) < 10.0


@pytest.mark.gpu
def test_reference_benchmark_setup_runs_unchanged(gpu):
    out = _run("benchmark")    # setup.cpp.bak:5-36: 256^3 D3Q19 SRT FP32, 1000 batches of 10 steps
    m = re.search(r"Peak MLUPs/s = (\d+)", out)
    assert m, out[-1000:]
    assert int(m.group(1)) > 25000
