// Drag on a sphere in creeping flow against Stokes' law (FluidX3D-master/src/setup.cpp.bak:148-187);
// build with -DFORCE_FIELD -DEQUILIBRIUM_BOUNDARIES.  Prints "force fx fy fz stokes F torque tx ty tz com cx cy cz".
#include "../setup.hpp"
void main_setup() {
	const float R = 16.0f, Re = 0.01f, nu = 1.0f, rho0 = 1.0f;
	const uint L = to_uint(8.0f*R);
	const float u = units.u_from_Re(Re, 2.0f*R, nu);
	const char* steps_env = std::getenv("LBM_EXAMPLE_STEPS");
	const ulong steps = steps_env ? (ulong)std::atol(steps_env) : 200ull;
	LBM lbm(L, L, L, nu);
	const uint Nx = lbm.get_Nx(), Ny = lbm.get_Ny(), Nz = lbm.get_Nz();
	const float3 c = lbm.center(), u0 = float3(-u, 0.0f, 0.0f);
	parallel_for(lbm.get_N(), [&](ulong n) {
		uint x = 0u, y = 0u, z = 0u; lbm.coordinates(n, x, y, z);
		if (x == 0u || x == Nx-1u || y == 0u || y == Ny-1u || z == 0u || z == Nz-1u) lbm.flags[n] = TYPE_E;
		const float3 p = lbm.position(x, y, z);
		if (length(float3((float)x, (float)y, (float)z)-c) <= R) { lbm.flags[n] = TYPE_S|TYPE_X; return; }
		lbm.rho[n] = units.rho_Stokes(p, u0, R, rho0, nu);
		const float3 v = units.u_Stokes(p, u0, R);
		lbm.u.x[n] = v.x; lbm.u.y[n] = v.y; lbm.u.z[n] = v.z;
	});
	lbm.run(steps);
	const float3 com = lbm.object_center_of_mass(TYPE_S|TYPE_X);
	const float3 force = lbm.object_force(TYPE_S|TYPE_X);
	const float3 torque = lbm.object_torque(com, TYPE_S|TYPE_X);
	println("force "+to_string(force.x, 8u)+" "+to_string(force.y, 8u)+" "+to_string(force.z, 8u)+" stokes "+to_string(units.F_Stokes(rho0, u, nu, R), 8u)
		+" torque "+to_string(torque.x, 8u)+" "+to_string(torque.y, 8u)+" "+to_string(torque.z, 8u)+" com "+to_string(com.x, 4u)+" "+to_string(com.y, 4u)+" "+to_string(com.z, 4u));
}
