// The reference's benchmark setup (FluidX3D-master/src/setup.cpp.bak:5-36): a box at rest, nu = 1, batches of 10 steps,
// peak MLUPs/s over the batches.  Build with -DD3Q19 -DSRT and optionally -DFP16S or -DFP16C.
// Environment: LBM_BENCH_N (cells per axis, default 512), LBM_BENCH_BATCHES (default 100), LBM_BENCH_DOMAINS ("DxxDyxDz",
// default 1x1x1; N is the size of ONE domain then, as in the reference's multi-GPU benchmark lines).
#include "../setup.hpp"
void main_setup() {
	const char* n_env = std::getenv("LBM_BENCH_N");
	const char* b_env = std::getenv("LBM_BENCH_BATCHES");
	const char* d_env = std::getenv("LBM_BENCH_DOMAINS");
	const uint N = n_env ? (uint)std::atoi(n_env) : 512u;
	const uint batches = b_env ? (uint)std::atoi(b_env) : 100u;
	uint Dx = 1u, Dy = 1u, Dz = 1u;
	if (d_env && std::sscanf(d_env, "%ux%ux%u", &Dx, &Dy, &Dz) != 3) print_error("LBM_BENCH_DOMAINS must look like 1x1x2");
	uint mlups = 0u;
	{
		LBM lbm(Dx*N, Dy*N, Dz*N, Dx, Dy, Dz, 1.0f);
		for (uint i = 0u; i < batches && running; i++) {
			lbm.run(10u, (ulong)batches*10ull);
			mlups = std::max(mlups, to_uint((double)lbm.get_N()*1E-6/info.runtime_lbm_timestep_smooth));
		}
	}
	print_info("Peak MLUPs/s = "+to_string(mlups));
	println("peak_mlups "+to_string(mlups));
}
