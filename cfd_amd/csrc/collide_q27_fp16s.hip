// stream_collide / update_fields instances for D3Q27 with FP16S DDF storage (gfx950).
#include "lbm_dispatch.hpp"
namespace lbm {
collide_launch_fn select_collide_q27_fp16s(int COLL, int FEAT, int V, int P, bool FO) {
	LBM_GENERIC(27, ST_FP16S)
	LBM_SPECIALISE(27, ST_FP16S, 0)
	LBM_SPECIALISE(27, ST_FP16S, F_VOLUME_FORCE|F_FORCE_FIELD)
	return nullptr;
}
} // namespace lbm
