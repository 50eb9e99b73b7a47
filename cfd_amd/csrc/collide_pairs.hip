// Instances of k_collide_pair and k_collide_band / k_collide_seam: two time steps of a single periodic domain, or of the inner
// planes of a z-slab, in one launch (gfx950).
#include "lbm_dispatch.hpp"
namespace lbm {
#define LBM_PAIR(Q_, ST_, COLL_, MASK, V_, P_) if (Q == (Q_) && ST == (ST_) && COLL == (COLL_) && FEAT == (int)(MASK) && V == (V_) && P == (P_)) return PairKernels{launch_collide_pair<Q_, ST_, COLL_, (int)(MASK), V_, P_>, launch_collide_band<Q_, ST_, COLL_, (int)(MASK), V_, P_>, launch_collide_seam<Q_, ST_, COLL_, (int)(MASK), V_, P_>};
PairKernels select_collide_pair(int Q, int ST, int COLL, int FEAT, int V, int P) {
	LBM_PAIR(19, ST_FP32, COLL_SRT, 0, 1, 1)                                   // benchmark: no extension
	LBM_PAIR(19, ST_FP32, COLL_TRT, 0, 1, 1)
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID, 1, 1)            // force-driven channel
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID, 1, 1)  // the reference's aerodynamics setups
	LBM_PAIR(27, ST_FP32, COLL_SRT, 0, 1, 1)
	LBM_PAIR(19, ST_FP16S, COLL_SRT, 0, 2, 1)                                  // 16-bit DDFs: two cells per lane (dword accesses)
	LBM_PAIR(19, ST_FP16S, COLL_SRT, 0, 2, 2)
	LBM_PAIR(19, ST_FP16C, COLL_SRT, 0, 2, 1)
	LBM_PAIR(19, ST_FP16S, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID, 2, 1)           // force-driven channel, 16-bit DDFs
	LBM_PAIR(19, ST_FP16S, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID, 2, 2)
	LBM_PAIR(27, ST_FP16C, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE, 2, 1)       // D3Q27 with force field (the 2048^3 configuration)
	LBM_PAIR(27, ST_FP16C, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE, 2, 2)
	LBM_PAIR(27, ST_FP16S, COLL_SRT, 0, 2, 1)
	LBM_PAIR(27, ST_FP16C, COLL_SRT, 0, 2, 1)
	return PairKernels{nullptr, nullptr, nullptr};
}
} // namespace lbm
