// stream_collide / update_fields instances for the small velocity sets D2Q9 and D3Q15 (run-time features only).
#include "lbm_dispatch.hpp"
namespace lbm {
template <int Q> static collide_launch_fn by_storage(int ST, int COLL, int FEAT, int V, int P, bool FO) {
	(void)V;
	if (FEAT >= 0) return nullptr;
	switch (ST) {
		case ST_FP32: if (FO) return COLL == COLL_TRT ? launch_collide<Q, ST_FP32, COLL_TRT, -1, true> : launch_collide<Q, ST_FP32, COLL_SRT, -1, true>;
			return COLL == COLL_TRT ? launch_collide<Q, ST_FP32, COLL_TRT, -1, false> : launch_collide<Q, ST_FP32, COLL_SRT, -1, false>;
		case ST_FP16S: if (FO) return COLL == COLL_TRT ? launch_collide<Q, ST_FP16S, COLL_TRT, -1, true> : launch_collide<Q, ST_FP16S, COLL_SRT, -1, true>;
			return COLL == COLL_TRT ? launch_collide<Q, ST_FP16S, COLL_TRT, -1, false> : launch_collide<Q, ST_FP16S, COLL_SRT, -1, false>;
		default: if (FO) return COLL == COLL_TRT ? launch_collide<Q, ST_FP16C, COLL_TRT, -1, true> : launch_collide<Q, ST_FP16C, COLL_SRT, -1, true>;
			return COLL == COLL_TRT ? launch_collide<Q, ST_FP16C, COLL_TRT, -1, false> : launch_collide<Q, ST_FP16C, COLL_SRT, -1, false>;
	}
}
collide_launch_fn select_collide_small(int Q, int ST, int COLL, int FEAT, int V, int P, bool FO) {
	return Q == 9 ? by_storage<9>(ST, COLL, FEAT, V, P, FO) : by_storage<15>(ST, COLL, FEAT, V, P, FO);
}
} // namespace lbm
