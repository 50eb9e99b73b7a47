// Instances of k_collide_pair and k_collide_band / k_collide_seam: two time steps of a single periodic domain, or of the inner
// planes of a z-slab, in one launch (gfx950).
#include "lbm_dispatch.hpp"
namespace lbm {
#define LBM_PAIR(Q_, ST_, COLL_, MASK, V_, P_, PAYS) if (Q == (Q_) && ST == (ST_) && COLL == (COLL_) && FEAT == (int)(MASK) && V == (V_) && P == (P_)) return PairKernels{launch_collide_pair<Q_, ST_, COLL_, (int)(MASK), V_, P_>, launch_collide_band<Q_, ST_, COLL_, (int)(MASK), V_, P_>, launch_collide_seam<Q_, ST_, COLL_, (int)(MASK), V_, P_>, PAYS};
// D3Q19 SRT FP32 has an instance for every combination of VOLUME_FORCE, EQUILIBRIUM_BOUNDARIES, FORCE_FIELD and SUBGRID: all of them gain.
// Last column: band by band against a launch per step, 512^3 (D3Q27: 384^3 too), vortex field, in % of 8 TB/s (tools/config_bench.py --fuse=2 --moving).
// The kernels that fill the vector issue slots already (16-bit DDFs with extensions) or hold few waves (132 registers) lose; instances
// that lost and have no other use were dropped: FP32 + MOVING_BOUNDARIES (55 against 66), + EQUILIBRIUM_BOUNDARIES + SUBGRID (54 against 68),
// FP16S + EQUILIBRIUM_BOUNDARIES + SUBGRID (58 against 65), FP16S + SUBGRID (62 against 66).
PairKernels select_collide_pair(int Q, int ST, int COLL, int FEAT, int V, int P) {
	LBM_PAIR(19, ST_FP32, COLL_SRT, 0, 1, 1, true)                                    // benchmark: no extension; 88 against 68
	LBM_PAIR(19, ST_FP32, COLL_TRT, 0, 1, 1, true)                                    // 87 against 65
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_VOLUME_FORCE, 1, 1, true)                       // 78 against 66
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_SUBGRID, 1, 1, true)                            // 77 against 66
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID, 1, 1, true)   // 76 against 66
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_EQUILIBRIUM_BOUNDARIES, 1, 1, true)             // 87 against 66
	LBM_PAIR(19, ST_FP32, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID, 1, 1, false)            // force-driven channel; 64 against 69
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID, 1, 1, true)   // the reference's aerodynamics setups; 86 against 66
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_SUBGRID, 1, 1, true)                            // 84 against 66
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_VOLUME_FORCE, 1, 1, true)                       // 88 against 65
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_VOLUME_FORCE|F_SUBGRID, 1, 1, true)             // 74 against 65
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_EQUILIBRIUM_BOUNDARIES, 1, 1, true)             // 89 against 65
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_EQUILIBRIUM_BOUNDARIES, 1, 1, true) // Stokes-drag style setups; 96 against 70 (165 B per cell)
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD, 1, 1, true)                          // 96 against 74
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_VOLUME_FORCE|F_EQUILIBRIUM_BOUNDARIES, 1, 1, true)                        // 86 against 66
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_VOLUME_FORCE|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID, 1, 1, true)              // 78 against 65
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_SUBGRID, 1, 1, true)                                        // 93 against 70
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE, 1, 1, true)                                   // 85 against 64
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE|F_SUBGRID, 1, 1, true)                         // 70 against 65
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID, 1, 1, true)               // 92 against 69
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE|F_EQUILIBRIUM_BOUNDARIES, 1, 1, true)          // 85 against 65
	LBM_PAIR(19, ST_FP32, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE|F_EQUILIBRIUM_BOUNDARIES|F_SUBGRID, 1, 1, true) // 76 against 65
	LBM_PAIR(27, ST_FP32, COLL_SRT, 0, 1, 1, true)                                    // 86 against 66
	LBM_PAIR(19, ST_FP16S, COLL_SRT, 0, 2, 1, true)                                   // 16-bit DDFs: two cells per lane (dword accesses); 79 against 65
	LBM_PAIR(19, ST_FP16S, COLL_SRT, 0, 2, 2, true)
	LBM_PAIR(19, ST_FP16S, COLL_SRT, F_EQUILIBRIUM_BOUNDARIES, 2, 1, true)            // 74 against 66
	LBM_PAIR(19, ST_FP16S, COLL_SRT, F_VOLUME_FORCE, 2, 1, true)                      // 67 against 65
	LBM_PAIR(19, ST_FP16C, COLL_SRT, 0, 2, 1, true)                                   // 71 against 65
	LBM_PAIR(19, ST_FP16S, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID, 2, 1, false)           // force-driven channel, 16-bit DDFs; 44-49 against 57
	LBM_PAIR(19, ST_FP16S, COLL_TRT, F_VOLUME_FORCE|F_SUBGRID, 2, 2, false)
	LBM_PAIR(27, ST_FP16C, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE, 2, 1, false)       // D3Q27 with force field (the 2048^3 configuration); 56-61 against 65
	LBM_PAIR(27, ST_FP16C, COLL_SRT, F_FORCE_FIELD|F_VOLUME_FORCE, 2, 2, false)
	LBM_PAIR(27, ST_FP16S, COLL_SRT, 0, 2, 1, true)                                   // 75 against 64
	LBM_PAIR(27, ST_FP16C, COLL_SRT, 0, 2, 1, true)                                   // 72 against 64
	return PairKernels{nullptr, nullptr, nullptr, false};
}
} // namespace lbm
