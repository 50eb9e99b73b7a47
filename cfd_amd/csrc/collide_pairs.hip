// Instances of k_collide_pair: two time steps of a single periodic domain in one launch (gfx950).
#include "lbm_dispatch.hpp"
namespace lbm {
#define LBM_PAIR(Q_, ST_, COLL_, MASK) if (Q == (Q_) && ST == (ST_) && COLL == (COLL_) && FEAT == (int)(MASK)) return launch_collide_pair<Q_, ST_, COLL_, (int)(MASK)>;
pair_launch_fn select_collide_pair(int Q, int ST, int COLL, int FEAT) {
	LBM_PAIR(19, ST_FP32, COLL_SRT, 0)                                   // benchmark: no extension
	LBM_PAIR(19, ST_FP32, COLL_TRT, 0)
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID)            // force-driven channel
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID)  // the reference's aerodynamics setups
	LBM_PAIR(27, ST_FP32, COLL_SRT, 0)
	return nullptr;
}
} // namespace lbm
