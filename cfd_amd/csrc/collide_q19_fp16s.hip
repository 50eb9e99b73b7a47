// stream_collide / update_fields instances for D3Q19 with FP16S DDF storage (gfx950).
#include "lbm_dispatch.hpp"
namespace lbm {
collide_launch_fn select_collide_q19_fp16s(int COLL, int FEAT, int V, int P, bool FO) {
	LBM_GENERIC(19, ST_FP16S)
	LBM_SPECIALISE(19, ST_FP16S, 0)                                                       // benchmark: no extension
	LBM_SPECIALISE(19, ST_FP16S, F_MOVING_BOUNDARIES)                                     // lid-driven cavity
	LBM_SPECIALISE(19, ST_FP16S, F_VOLUME_FORCE|F_SUBGRID)                                // force-driven channel
	LBM_SPECIALISE(19, ST_FP16S, F_UPDATE_FIELDS|F_VOLUME_FORCE|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID) // the GUI's shipped defines.hpp
	LBM_SPECIALISE(19, ST_FP16S, F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID)                           // the reference's aerodynamics setups (setup.cpp.bak: 8 of 24)
	LBM_SPECIALISE(19, ST_FP16S, F_MOVING_BOUNDARIES|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID)       // ... with rotating parts
	LBM_SPECIALISE(19, ST_FP16S, F_MOVING_BOUNDARIES|F_SUBGRID)                             // setup.cpp.bak: rotating geometry without inflow
	return nullptr;
}
} // namespace lbm
