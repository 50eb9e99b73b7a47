// What the GUI generates (main.py:145-198) with its placeholders filled in: resolution from aspect ratio and VRAM,
// STL voxelized into the box, inflow velocity everywhere outside the body, TYPE_E on all six faces, then a loop that
// honours run/pause (key_P), export (key_9) and stop (running).  Build with the GUI's defines:
// -DD3Q19 -DSRT -DFP16S -DEQUILIBRIUM_BOUNDARIES -DSUBGRID -DVOLUME_FORCE -DUPDATE_FIELDS
#include "../setup.hpp"
void main_setup() {
	const char* stl_env = std::getenv("LBM_EXAMPLE_STL");
	const char* steps_env = std::getenv("LBM_EXAMPLE_STEPS");
	const uint3 lbm_N = resolution(float3(1.0f, 1.0f, 1.0f), 24u); // aspect ratio and MB of VRAM -> grid
	const float lbm_Re = 1000.0f, lbm_u = 0.075f;
	const ulong lbm_T = steps_env ? (ulong)std::atol(steps_env) : 200ull;
	LBM lbm(lbm_N, 1u, 1u, 1u, units.nu_from_Re(lbm_Re, (float)lbm_N.x, lbm_u));
	const float size = 0.5f*lbm.size().z;
	const float3 center = float3(lbm.center().x, lbm.center().y, lbm.center().z);
	const float3x3 rotation = float3x3(float3(1, 0, 0), radians(0.0f))*float3x3(float3(0, 1, 0), radians(0.0f))*float3x3(float3(0, 0, 1), radians(30.0f));
	Clock clock;
	lbm.voxelize_stl(stl_env ? string(stl_env) : get_exe_path()+"../stl/body.stl", center, rotation, size);
	println(print_time(clock.stop()));
	const uint Nx = lbm.get_Nx(), Ny = lbm.get_Ny(), Nz = lbm.get_Nz();
	parallel_for(lbm.get_N(), [&](ulong n) {
		uint x = 0u, y = 0u, z = 0u; lbm.coordinates(n, x, y, z);
		if (lbm.flags[n] != TYPE_S) lbm.u.x[n] = lbm_u;
		if (x == 0u || x == Nx-1u || y == 0u || y == Ny-1u || z == 0u || z == Nz-1u) lbm.flags[n] = TYPE_E;
	});
	lbm.graphics.visualization_modes = VIS_FLAG_LATTICE|VIS_FLAG_SURFACE|VIS_Q_CRITERION;
	const string export_path = get_exe_path()+"export/";
	lbm.write_status(export_path);
	{ // the body as the reference would export it beside the fields
		Mesh* mesh = read_stl(stl_env ? string(stl_env) : get_exe_path()+"../stl/body.stl", lbm.size(), center, rotation, size);
		lbm.write_mesh_to_vtk(mesh, export_path);
		delete mesh;
	}
	lbm.run(0u, lbm_T);
	key_9 = true; // one export, as if the GUI had pressed the key
	while (lbm.get_t() < lbm_T && running) {
		if (key_9) {
			lbm.u.write_device_to_vtk(export_path);
			lbm.rho.write_device_to_vtk(export_path);
			lbm.flags.write_device_to_vtk(export_path);
			key_9 = false;
		}
		if (!key_P) { sleep(0.016); continue; }
		lbm.run(20u, lbm_T);
	}
	lbm.write_status(export_path);
	lbm.u.read_from_device();
	ulong solid = 0ull;
	for (ulong n = 0ull; n < lbm.get_N(); n++) solid += lbm.flags[n] == TYPE_S;
	println("grid "+to_string(Nx)+"x"+to_string(Ny)+"x"+to_string(Nz)+" solid "+to_string(solid)+" t "+to_string(lbm.get_t()));
}
