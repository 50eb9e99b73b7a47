// Misconfigured setups, one per value of the environment variable LBM_ERROR_CASE: each must end like the reference
// (FluidX3D-master/src/utilities.hpp:4159-4166, lbm.cpp:1038-1124, 1354-1409): an "| Error:" line and exit status 1.
// Built as D3Q19 SRT without any extension.
#include "../setup.hpp"
#include <cstdlib>

void main_setup() {
	const char* env = std::getenv("LBM_ERROR_CASE");
	const int which = env ? std::atoi(env) : 0;
	switch (which) {
		case 1: { LBM lbm(32u, 32u, 32u, 0.0f); lbm.run(1u); break; }                       // viscosity 0
		case 2: { LBM lbm(32u, 32u, 32u, -0.1f); lbm.run(1u); break; }                      // negative viscosity
		case 3: { LBM lbm(32u, 32u, 32u, 0.1f, 0.001f, 0.0f, 0.0f); lbm.run(1u); break; }   // volume force without VOLUME_FORCE
		case 4: { LBM lbm(32u, 32u, 32u, 0.1f); lbm.flags[5] = TYPE_E; lbm.run(1u); break; } // TYPE_E without EQUILIBRIUM_BOUNDARIES
		case 5: { LBM lbm(0u, 32u, 32u, 0.1f); lbm.run(1u); break; }                        // empty lattice
		case 6: { LBM lbm(32u, 32u, 32u, 0.1f); lbm.voxelize_stl("/nonexistent/mesh.stl", 16.0f); lbm.run(1u); break; } // missing geometry file
		case 7: { LBM lbm(32u, 32u, 32u, 0u, 1u, 1u, 0.1f); lbm.run(1u); break; }           // 0 domains along x
		default: { LBM lbm(32u, 32u, 32u, 0.1f); lbm.run(1u); println("no error case selected: ran 1 step"); break; }
	}
}
