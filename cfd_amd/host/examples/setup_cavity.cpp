// Lid-driven cavity (FluidX3D-master/src/setup.cpp.bak:236-249); build with -DMOVING_BOUNDARIES.
#include "../setup.hpp"
void main_setup() {
	const uint L = 128u;
	const float Re = 1000.0f, u = 0.1f;
	LBM lbm(L, L, L, units.nu_from_Re(Re, (float)(L-2u), u));
	const uint Nx = lbm.get_Nx(), Ny = lbm.get_Ny(), Nz = lbm.get_Nz();
	parallel_for(lbm.get_N(), [&](ulong n) {
		uint x = 0u, y = 0u, z = 0u; lbm.coordinates(n, x, y, z);
		if (z == Nz-1u) lbm.u.y[n] = u;
		if (x == 0u || x == Nx-1u || y == 0u || y == Ny-1u || z == 0u || z == Nz-1u) lbm.flags[n] = TYPE_S;
	});
	lbm.run(1000u);
	lbm.u.read_from_device();
	println(to_string(lbm.u.y[lbm.index(Nx/2u, Ny/2u, Nz-2u)]));
	print_info("MLUPs/s = "+to_string((float)((double)lbm.get_N()*1E-6/info.runtime_lbm_timestep_smooth), 0u));
}
