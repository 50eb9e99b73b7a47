// fluid_wrapper.cpp - libfluid_wrapper.so: the C entry points solver.py loads (see include/fluid_wrapper.h).
// Compiled with -DD3Q19 -DSRT -DVOLUME_FORCE -DUPDATE_FIELDS together with lbm.cpp.
#include "../../include/fluid_wrapper.h"
#include "lbm.hpp"

static LBM* g_lbm = nullptr;

extern "C" void fluid_cleanup(void) {
	delete g_lbm;
	g_lbm = nullptr;
}
extern "C" void fluid_init(int resolution, float viscosity, float pump_force, const char* stl_path) {
	fluid_cleanup();
	const uint N = (uint)std::max(resolution, 4);
	g_lbm = new LBM(N, N, N, viscosity, 0.0f, 0.0f, pump_force);
	if (stl_path && stl_path[0]) g_lbm->voxelize_stl(string(stl_path), 0.9f*(float)N); // longest side = 90 % of the box
	g_lbm->run(0u); // uploads the fields and initializes the DDFs
}
extern "C" void fluid_step(int steps) {
	if (g_lbm && steps > 0) g_lbm->run((ulong)steps);
}
extern "C" void fluid_get_velocity(float* vx, float* vy, float* vz) {
	if (!g_lbm || !vx || !vy || !vz) return;
	g_lbm->u.read_from_device();
	const ulong N = g_lbm->get_N();
	parallel_for(N, [&](ulong n) { vx[n] = g_lbm->u.x[n]; vy[n] = g_lbm->u.y[n]; vz[n] = g_lbm->u.z[n]; });
}
