// stream_collide / update_fields instances for D3Q19 with FP32 DDF storage (gfx950).
#include "lbm_dispatch.hpp"
namespace lbm {
collide_launch_fn select_collide_q19_fp32(int COLL, int FEAT, int V, int P, bool FO) {
	LBM_GENERIC(19, ST_FP32)
	LBM_SPECIALISE(19, ST_FP32, 0)                                                       // benchmark: no extension
	LBM_SPECIALISE(19, ST_FP32, F_MOVING_BOUNDARIES)                                     // lid-driven cavity
	LBM_SPECIALISE(19, ST_FP32, F_VOLUME_FORCE|F_SUBGRID)                                // force-driven channel
	LBM_SPECIALISE(19, ST_FP32, F_UPDATE_FIELDS|F_VOLUME_FORCE|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID) // the GUI's shipped defines.hpp
	LBM_SPECIALISE(19, ST_FP32, F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID)                           // the reference's aerodynamics setups (setup.cpp.bak: 8 of 24)
	LBM_SPECIALISE(19, ST_FP32, F_MOVING_BOUNDARIES|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID)       // ... with rotating parts
	return nullptr;
}
} // namespace lbm
