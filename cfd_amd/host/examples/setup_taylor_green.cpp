// 3D Taylor-Green vortices, the setup of FluidX3D-master/src/setup.cpp.bak:40-59 written against this host; prints u.x at
// the box centre after 1000 steps (the reference's "test for binary identity" probe) and the measured MLUPs/s.
#include "../setup.hpp"
// LBM_EXAMPLE_DOMAINS="DxxDyxDz" splits the lattice (default 1x1x1); the checksum line is the same for every split.
void main_setup() {
	uint Dx = 1u, Dy = 1u, Dz = 1u;
	if (const char* d_env = std::getenv("LBM_EXAMPLE_DOMAINS")) if (std::sscanf(d_env, "%ux%ux%u", &Dx, &Dy, &Dz) != 3) print_error("LBM_EXAMPLE_DOMAINS must look like 1x1x2");
	LBM lbm(128u, 128u, 128u, Dx, Dy, Dz, 0.01f);
	const uint Nx = lbm.get_Nx(), Ny = lbm.get_Ny(), Nz = lbm.get_Nz();
	parallel_for(lbm.get_N(), [&](ulong n) {
		uint x = 0u, y = 0u, z = 0u; lbm.coordinates(n, x, y, z);
		const float A = 0.25f;
		const float a = (float)Nx, b = (float)Ny, c = (float)Nz;
		const float fx = (float)x+0.5f-0.5f*(float)Nx, fy = (float)y+0.5f-0.5f*(float)Ny, fz = (float)z+0.5f-0.5f*(float)Nz;
		lbm.u.x[n] =  A*cosf(2.0f*pif*fx/a)*sinf(2.0f*pif*fy/b)*sinf(2.0f*pif*fz/c);
		lbm.u.y[n] = -A*sinf(2.0f*pif*fx/a)*cosf(2.0f*pif*fy/b)*sinf(2.0f*pif*fz/c);
		lbm.u.z[n] =  A*sinf(2.0f*pif*fx/a)*sinf(2.0f*pif*fy/b)*cosf(2.0f*pif*fz/c);
		lbm.rho[n] = 1.0f-sq(A)*3.0f/4.0f*(cosf(4.0f*pif*fx/a)+cosf(4.0f*pif*fy/b));
	});
	const char* steps_env = std::getenv("LBM_EXAMPLE_STEPS"); // default: the reference's 1000-step probe
	lbm.run(steps_env ? (ulong)std::atol(steps_env) : 1000ull);
	lbm.u.read_from_device();
	println(to_string(lbm.u.x[lbm.index(Nx/2u, Ny/2u, Nz/2u)]));
	ulong checksum = 0ull; // of the raw bits of the whole velocity field
	for (ulong n = 0ull; n < lbm.get_N(); n++) {
		const float v[3] = {lbm.u.x[n], lbm.u.y[n], lbm.u.z[n]};
		for (int c = 0; c < 3; c++) { uint bits = 0u; std::memcpy(&bits, &v[c], 4); checksum = checksum*1099511628211ull+(ulong)bits; }
	}
	println("u checksum "+to_string(checksum));
	print_info("MLUPs/s = "+to_string((float)((double)lbm.get_N()*1E-6/info.runtime_lbm_timestep_smooth), 0u));
	lbm.u.write_host_to_vtk(get_exe_path()+"export/");
	lbm.write_status(get_exe_path()+"export/");
}
