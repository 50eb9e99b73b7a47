#!/bin/bash
# Diagnostic: builds variants of the library with k_collide_band's experiment switches (LBM_BAND_EXP, lbm_kernels.hpp) into tools/ab/.
# Only the pair / band instances differ; the other objects are linked from the regular build.
# usage: tools/band_variants.sh 1 2 4 ...   ->  tools/ab/lbm_exp<N>.so
set -e
cd "$(dirname "$0")/../cfd_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/ab
for n in "$@"; do
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DLBM_BAND_EXP=$n -c collide_pairs.hip -o build/collide_pairs_exp$n.o
	objs=$(ls build/*.o | grep -v collide_pairs)
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lbm_exp$n.so $objs build/collide_pairs_exp$n.o -ldl
	echo "tools/ab/lbm_exp$n.so"
done
