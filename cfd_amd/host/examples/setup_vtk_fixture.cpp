// Writes every field of a 4 x 3 x 2 lattice with known values as legacy VTK, plus the status file, into the directory named
// by LBM_EXAMPLE_OUT.  The expected bytes are worked out from the reference's file format alone (tests/golden/make_vtk_golden.py:
// lbm.hpp:298-340 header and payload, lbm.cpp:104-118 file names, lbm.cpp:1599-1643 status keys, units.hpp:9-13, 58-71 scaling).
#include "../setup.hpp"
void main_setup() {
	units.set_m_kg_s(1.0f, 1.0f, 1.0f, 2.0f, 4.0f, 8.0f); // one cell = 2 m, unit velocity = 4 m/s, unit density = 8 kg/m^3
	LBM lbm(4u, 3u, 2u, 0.125f, 0.0f, 0.0f, 0.0009765625f);
	for (ulong n = 0ull; n < lbm.get_N(); n++) { // all values are exact binary fractions, so the scaled values are exact too
		const float v = (float)n;
		lbm.rho[n] = 1.0f+v/64.0f;
		lbm.u.x[n] = v/128.0f; lbm.u.y[n] = -v/256.0f; lbm.u.z[n] = v/512.0f;
		lbm.flags[n] = (uchar)((n*7ull)%256ull);
		lbm.F.x[n] = v/16.0f; lbm.F.y[n] = -v/32.0f; lbm.F.z[n] = v/8.0f;
	}
	println("pinned "+to_string((uint)lbm.lbm_domain[0]->rho.is_pinned())); // host copies in page-locked memory unless LBM_PINNED=0
	const char* out = std::getenv("LBM_EXAMPLE_OUT");
	const string path = out ? string(out) : get_exe_path()+"export/";
	lbm.rho.write_host_to_vtk(path);
	lbm.u.write_host_to_vtk(path);
	lbm.flags.write_host_to_vtk(path);
	lbm.F.write_host_to_vtk(path);
	lbm.write_status(path);
}
