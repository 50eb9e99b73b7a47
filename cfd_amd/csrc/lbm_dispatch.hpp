// lbm_dispatch.hpp - run-time selection of a compiled stream_collide / update_fields instance.
// Each (velocity set, storage) pair lives in its own translation unit so that the build parallelises.
#pragma once
#include "lbm_kernels.hpp"

namespace lbm {

typedef hipError_t (*collide_launch_fn)(const Args&, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream);

template <int Q, int ST, int COLL, int FEAT, bool FO>
hipError_t launch_collide(const Args& a, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream) {
	hipLaunchKernelGGL((k_collide<Q, ST, COLL, FEAT, FO>), grid, block, lds_bytes, stream, a);
	return hipGetLastError();
}

template <int Q, int ST, int COLL, int FEAT, int V, int P>
hipError_t launch_collide_wide(const Args& a, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream) {
	hipLaunchKernelGGL((k_collide_wide<Q, ST, COLL, FEAT, V, P>), grid, block, lds_bytes, stream, a);
	return hipGetLastError();
}

template <int Q, int ST, int V>
hipError_t launch_fields_wide(const Args& a, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream) { // update_fields, V cells per lane
	hipLaunchKernelGGL((k_collide_wide<Q, ST, COLL_SRT, -1, V, 1, true>), grid, block, lds_bytes, stream, a);
	return hipGetLastError();
}

// two time steps in one launch (k_collide_pair); nullptr when the configuration has no compiled instance
typedef hipError_t (*pair_launch_fn)(const Args&, const PairArgs&, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream);
template <int Q, int ST, int COLL, int FEAT, int V, int P>
hipError_t launch_collide_pair(const Args& a, const PairArgs& p, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream) {
	hipLaunchKernelGGL((k_collide_pair<Q, ST, COLL, FEAT, V, P>), grid, block, lds_bytes, stream, a, p);
	return hipGetLastError();
}
// the same two steps band by band on each XCD (k_collide_band) and the launch that finishes them (k_collide_seam)
typedef hipError_t (*band_launch_fn)(const Args&, const BandArgs&, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream);
typedef hipError_t (*seam_launch_fn)(const Args&, const SeamArgs&, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream);
template <int Q, int ST, int COLL, int FEAT, int V, int P>
hipError_t launch_collide_band(const Args& a, const BandArgs& p, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream) {
	hipLaunchKernelGGL((k_collide_band<Q, ST, COLL, FEAT, V, P>), grid, block, lds_bytes, stream, a, p);
	return hipGetLastError();
}
template <int Q, int ST, int COLL, int FEAT, int V, int P>
hipError_t launch_collide_seam(const Args& a, const SeamArgs& p, dim3 grid, dim3 block, uint32_t lds_bytes, hipStream_t stream) {
	hipLaunchKernelGGL((k_collide_seam<Q, ST, COLL, FEAT, V, P>), grid, block, lds_bytes, stream, a, p);
	return hipGetLastError();
}
struct PairKernels { pair_launch_fn pair; band_launch_fn band; seam_launch_fn seam; bool pays; }; // pays: measured faster than a launch per step on MI355X (512^3, moving field) - what automatic mode goes by
PairKernels select_collide_pair(int Q, int ST, int COLL, int FEAT, int V, int P); // V = 1 (FP32 DDFs) or 2 (16-bit DDFs); all null without an instance

// FEAT < 0 selects the instance that reads the feature mask at run time; FO = update_fields (always run-time features)
// V = cells per lane; P = 2 selects the instance that collides two cells at a time in packed arithmetic (V = 2 or 4 only)
template <int Q, int ST, int COLL, int FEAT> collide_launch_fn pick_v(int V, int P) {
	switch (V) {
		case 4: return P == 2 ? launch_collide_wide<Q, ST, COLL, FEAT, 4, 2> : launch_collide_wide<Q, ST, COLL, FEAT, 4, 1>;
		case 2: return P == 2 ? launch_collide_wide<Q, ST, COLL, FEAT, 2, 2> : launch_collide_wide<Q, ST, COLL, FEAT, 2, 1>;
		default: return launch_collide<Q, ST, COLL, FEAT, false>;
	}
}

// per translation unit: returns nullptr when (COLL, FEAT) has no compiled instance
collide_launch_fn select_collide_q19_fp32(int COLL, int FEAT, int V, int P, bool FO);
collide_launch_fn select_collide_q19_fp16s(int COLL, int FEAT, int V, int P, bool FO);
collide_launch_fn select_collide_q19_fp16c(int COLL, int FEAT, int V, int P, bool FO);
collide_launch_fn select_collide_q27_fp32(int COLL, int FEAT, int V, int P, bool FO);
collide_launch_fn select_collide_q27_fp16s(int COLL, int FEAT, int V, int P, bool FO);
collide_launch_fn select_collide_q27_fp16c(int COLL, int FEAT, int V, int P, bool FO);
collide_launch_fn select_collide_small(int Q, int ST, int COLL, int FEAT, int V, int P, bool FO); // D2Q9, D3Q15

// LBM_SELECT_BODY(Q, ST, specialised feature masks...) expands to the body of one select_collide_* function
#define LBM_SPECIALISE(Q, ST, MASK) \
	if (FEAT == (MASK)) return COLL == COLL_TRT ? pick_v<Q, ST, COLL_TRT, (MASK)>(V, P) : pick_v<Q, ST, COLL_SRT, (MASK)>(V, P);
#define LBM_GENERIC(Q, ST) \
	if (FO) return V == 4 ? launch_fields_wide<Q, ST, 4> : V == 2 ? launch_fields_wide<Q, ST, 2> : launch_collide<Q, ST, COLL_SRT, -1, true>; /* the moments do not depend on the collision operator */ \
	if (FEAT < 0) return COLL == COLL_TRT ? pick_v<Q, ST, COLL_TRT, -1>(V, P) : pick_v<Q, ST, COLL_SRT, -1>(V, P);

} // namespace lbm
