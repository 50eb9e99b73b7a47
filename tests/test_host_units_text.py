"""CPU: unit conversion and text formats of the C++ host (cfd_amd/host/units.hpp, utilities.hpp).

Unit conversion (FluidX3D-master/src/units.hpp:8-40): set_m_kg_s(x, u, rho, si_x, si_u, si_rho) fixes metre, kilogram and
second per lattice unit; every other quantity follows from its SI dimension.  The expected values below are computed in
Python from the dimensions alone.  Text: to_string(float) is "d.dddddddd[E<exp>]" (utilities.hpp:2745-2765) and
print_time drops leading zero units (utilities.hpp:2876-2888)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = r'''
#include "units.hpp"
#include <cstdio>
Units units;
int main() {
	units.set_m_kg_s(128.0f, 0.1f, 1.0f, 2.0f, 10.0f, 1.225f); // 128 cells = 2 m, lattice 0.1 = 10 m/s, lattice density 1 = 1.225 kg/m^3
	std::printf("%.9g %.9g %.9g\n", units.si_x(1.0f), units.si_m(1.0f), units.si_t(1ull));
	std::printf("%.9g %.9g %llu %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", units.x(0.5f), units.m(3.0f), (unsigned long long)units.t(0.25f), units.u(4.0f), units.rho(1000.0f),
		units.nu(1.48E-5f), units.mu(1.8E-5f), units.g(9.81f), units.f(12.0f), units.F(2.5f), units.M(0.75f), units.sigma(0.072f));
	std::printf("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", units.si_x(64.0f), units.si_t(1000ull), units.si_u(0.05f), units.si_rho(1.1f), units.si_p(0.01f),
		units.si_nu(0.02f), units.si_F(1.5f), units.si_M(8.0f), units.si_f(1E-4f));
	std::printf("%.9g %.9g %.9g %.9g\n", units.u_from_Re(320.0f, 64.0f, 0.02f), units.si_Re(2.0f, 10.0f, 1.48E-5f), units.nu_from_Re(1000.0f, 64.0f, 0.1f), units.nu_from_tau(0.56f));
	const float v[] = {0.0f, 1.0f, -2.5f, 1234.5f, 0.000123f, 1.0E10f, -9.99999994E-1f, 3.0E-38f};
	for (const float x : v) std::printf("%s\n", to_string(x).c_str());
	std::printf("%s %s %s\n", to_string(0.1).c_str(), to_string(3.14159f, 3u).c_str(), to_string(42u).c_str());
	const double s[] = {0.2, 7.0, 65.0, 3600.0, 7387.0, 90061.0, 31556925.0+86400.0+61.0};
	for (const double x : s) std::printf("%s\n", print_time(x).c_str());
	return 0;
}
'''


def test_units_and_text(tmp_path):
    src = tmp_path / "units_harness.cpp"
    src.write_text(HARNESS)
    exe = str(tmp_path / "units_harness")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "cfd_amd", "host"), str(src), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()
    out = [l for l in out if not l.startswith("|")]  # the conversion report line
    m = 2.0 / 128.0                       # metres per cell
    s = m * 0.1 / 10.0                    # seconds per step: u_lattice = u_si * s / m
    kg = 1.225 * m ** 3 / 1.0             # kilograms per lattice mass unit: rho_lattice = rho_si * m^3 / kg
    base = [float(v) for v in out[0].split()]
    assert np.allclose(base, [m, kg, s], rtol=1e-6)
    to_lattice = lambda v, L, M, T: v / (m ** L * kg ** M * s ** T)
    got = out[1].split()
    want = [to_lattice(0.5, 1, 0, 0), to_lattice(3.0, 0, 1, 0), None, to_lattice(4.0, 1, 0, -1), to_lattice(1000.0, -3, 1, 0), to_lattice(1.48e-5, 2, 0, -1),
            to_lattice(1.8e-5, -1, 1, -1), to_lattice(9.81, 1, 0, -2), to_lattice(12.0, -2, 1, -2), to_lattice(2.5, 1, 1, -2), to_lattice(0.75, 2, 1, -2), to_lattice(0.072, 0, 1, -2)]
    assert int(got[2]) == int(0.25 / s)   # time in whole steps
    for g, w in zip(got, want):
        if w is not None:
            assert abs(float(g) - w) <= 2e-6 * abs(w), (g, w)
    to_si = lambda v, L, M, T: v * (m ** L * kg ** M * s ** T)
    want = [to_si(64.0, 1, 0, 0), to_si(1000.0, 0, 0, 1), to_si(0.05, 1, 0, -1), to_si(1.1, -3, 1, 0), to_si(0.01, -1, 1, -2), to_si(0.02, 2, 0, -1),
            to_si(1.5, 1, 1, -2), to_si(8.0, 2, 1, -2), to_si(1e-4, -2, 1, -2)]
    for g, w in zip(out[2].split(), want):
        assert abs(float(g) - w) <= 2e-6 * abs(w), (g, w)
    assert np.allclose([float(v) for v in out[3].split()], [320.0 * 0.02 / 64.0, 2.0 * 10.0 / 1.48e-5, 64.0 * 0.1 / 1000.0, 0.06 / 3.0], rtol=2e-6)
    # floats as the reference prints them: its own digit extraction in float arithmetic (utilities.hpp:2599-2630, 2745-2754,
    # restated independently in tests/golden/make_vtk_golden.py), whose last digits differ from a correctly rounded conversion
    from tests.golden.make_vtk_golden import ref_float_string
    values = [0.0, 1.0, -2.5, 1234.5, 0.000123, 1.0e10, -9.99999994e-1, 3.0e-38]
    assert out[4:12] == [ref_float_string(np.float32(v)) for v in values]
    assert out[4:8] == ["0.00000000", "1.00000000", "-2.50000000", "1.23449994E3"] and out[9] == "1.00000000E10"
    assert out[12] == "1.0000000000000000E-1 3.142 42"
    assert out[13:20] == ["0s", "7s", "1m05s", "1h00m00s", "2h03m07s", "1d01h01m01s", "1y001d00h01m01s"]
