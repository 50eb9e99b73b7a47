// lbm_kernels.hpp - the lattice kernels (templates).
//
// Thread mapping of the fused stream+collide kernels: one workgroup works on one lattice row (fixed y,z),
// or on an x-chunk of it; y, z and every neighbour-row base are therefore wave-uniform (SGPRs) and only
// the x offset is per lane.  k_collide gives every lane one cell (4- or 2-byte accesses); k_collide_wide
// gives it 2 or 4 consecutive cells moved as aligned vectors.  Every DDF slot is read and written by exactly
// one cell per step (esoteric-pull, FluidX3D-master/src/kernel.cpp:1332-1345), so there is no inter-thread
// hazard and the traffic is the algorithmic 2*Q*sizeof(fpxx) bytes per cell.
#pragma once
#include "lbm_device.hpp"

namespace lbm {

// wall velocity terms of a cell next to moving solids (kernel.cpp:1111-1118): every pair looks at both neighbours
template <int Q> __device__ __forceinline__ void apply_moving_boundaries(float* f, const Args& a, const uint32_t x, const uint32_t* yq, const uint32_t* zq) {
	const uint32_t xq[3] = {x, x+1u == a.Nx ? 0u : x+1u, x == 0u ? a.Nx-1u : x-1u}; // index 0: same, 1: +1, 2: -1
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I;
		constexpr float w6 = -6.0f*weight<Q>(i);
		auto nb = [&](auto J) -> uint64_t {
			constexpr int jx = dir<Q>(J,0), jy = dir<Q>(J,1), jz = dir<Q>(J,2);
			return ((uint64_t)zq[jz == 0 ? 0 : jz > 0 ? 1 : 2]*a.Ny+yq[jy == 0 ? 0 : jy > 0 ? 1 : 2])*a.Nx+xq[jx == 0 ? 0 : jx > 0 ? 1 : 2];
		};
		uint64_t ji = nb(std::integral_constant<int, i+1>{});
		if ((a.flags[ji]&TYPE_BO) == TYPE_S) f[i] = fmaf(w6, cdot<Q, i+1>(a.u[ji], a.u[a.N+ji], a.u[2ull*a.N+ji]), f[i]);
		ji = nb(std::integral_constant<int, i>{});
		if ((a.flags[ji]&TYPE_BO) == TYPE_S) f[i+1] = fmaf(w6, cdot<Q, i>(a.u[ji], a.u[a.N+ji], a.u[2ull*a.N+ji]), f[i+1]);
	});
}

// ---------------------------------------------------------------------------------------------- stream_collide / update_fields, one cell per lane
// FluidX3D-master/src/kernel.cpp:1470-1654 (FIELDS_ONLY=false) and 1812-1894 (FIELDS_ONLY=true).
// FEAT >= 0: features fixed at compile time; FEAT < 0: taken from Args::features.
// collide_row is the work of one workgroup: the x-chunk `chunk` of lattice row (y, z).  MEM chooses how the DDFs move
// (plain accesses, or the agent-scope forms k_collide_pair needs); flags, rho, u and F always move plainly.
// HAVE_FLAG: the caller has loaded the lane's flag byte already (`flag`).
template <int Q, int ST, int COLL, int FEAT, bool FIELDS_ONLY, int MEM = MEM_PLAIN, bool HAVE_FLAG = false>
__device__ __forceinline__ void collide_row(const Args& a, const uint32_t chunk, const uint32_t y, const uint32_t z, const uint32_t flag = 0u) {
	using S = Store<ST>;
	using raw = typename S::raw;
	if (a.hy && (y == 0u || y == a.Ny-1u)) return;
	const uint32_t x = chunk*blockDim.x+threadIdx.x;
	if (x >= a.Nx) return;

	// wave-uniform row bases of the 3x3 neighbourhood in (y,z), periodic on the local lattice (kernel.cpp:914-925)
	const uint32_t yq[3] = {y, y+1u == a.Ny ? 0u : y+1u, y == 0u ? a.Ny-1u : y-1u}; // index 0: same, 1: +1, 2: -1
	const uint32_t zq[3] = {z, z+1u == a.Nz ? 0u : z+1u, z == 0u ? a.Nz-1u : z-1u};
	const uint64_t row0 = ((uint64_t)z*a.Ny+y)*a.Nx;
	raw* const fi = static_cast<raw*>(a.fi);

	// solid cells and halo cells are skipped (kernel.cpp:1482-1485).  A solid cell in a wave that also holds fluid cells
	// stays to write the bits it owns back unchanged (DDF slots, and rho/u where those are written): memory sectors at
	// obstacle surfaces are then written whole, which costs far less than the partial writes left by lanes that drop out.
	// Waves made of skipped cells only, and the halo columns, leave at once.
	const uint8_t flagsn = HAVE_FLAG ? (uint8_t)flag : ld<uint8_t>(a.flags+row0, x), bo = flagsn&TYPE_BO;
	const bool halo_x = a.hx && (x == 0u || x == a.Nx-1u), skip = bo == TYPE_S || halo_x;
	if (__ballot(!skip) == 0ull || halo_x) return;
	auto feat = [&]() { if constexpr (FEAT >= 0) return StaticFeat<(uint32_t)(FEAT >= 0 ? FEAT : 0)>{}; else return DynamicFeat{a.features}; }();

	// ---- streaming part 2: gather (kernel.cpp:1332-1338); the +x / -x neighbour column, periodic
	const uint32_t xp = x+1u == a.Nx ? 0u : x+1u, xm = x == 0u ? a.Nx-1u : x-1u;
	constexpr uint32_t RB = sizeof(raw);
	const uint32_t xb = x*RB, xpb = xp*RB, xmb = xm*RB; // byte offsets inside a row
	raw r[Q]; // the stored bits: r[i] from the cell's own slot of pair (i, i+1), r[i+1] from the neighbour's
	// wave-uniform row bases: pa[0] and pa[i] the cell's own row in the streams it owns, pa[i+1] the row of its +c_i neighbours.
	// They are formed before the first load on purpose: with the scalar arithmetic out of the way the loads issue back to
	// back, which is worth 7 % of the step rate of the plain FP32 kernel against bases formed between the loads.
	raw* pa[Q];
	pa[0] = fi+row0;
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		const uint32_t sA = a.odd ? i : i+1, sB = a.odd ? i+1 : i;
		pa[i] = fi+(uint64_t)sA*a.S+row0;
		pa[i+1] = fi+(uint64_t)sB*a.S+((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx;
	});
	if (!FIELDS_ONLY || !skip) {
		if constexpr (MEM != MEM_LOAD_COHERENT) r[0] = ld_ddf<raw, MEM>(pa[0], xb);
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I, cx = dir<Q>(i,0);
			r[i] = ld_ddf<raw, MEM>(pa[i], xb);
			r[i+1] = ld_ddf<raw, MEM>(pa[i+1], cx == 0 ? xb : cx > 0 ? xpb : xmb);
		});
		if constexpr (MEM == MEM_LOAD_COHERENT) r[0] = ld_ddf<raw, MEM>(pa[0], xb); // last: the compiler waits for this one at once (it keeps a second copy of the value)
	}

	// ---- boundaries, moments, collision
	float rhon, uxn, uyn, uzn;
	bool write_fields = false;
	if (!skip) {
		float f[Q];
		static_for<0, Q, 1>([&](auto I) { f[I] = S::decode(r[I]); });
		if (feat(F_MOVING_BOUNDARIES) && bo == TYPE_MS) apply_moving_boundaries<Q>(f, a, x, yq, zq);
		collide<Q, COLL, FIELDS_ONLY>(f, feat, &bo, a, row0, x, rhon, uxn, uyn, uzn, &write_fields);
		if constexpr (!FIELDS_ONLY) { // streaming part 1 in registers: the slot that held f[i] receives f[i+1] and vice versa (kernel.cpp:1339-1345)
			S::template prescale<Q>(f);
			r[0] = S::encode_scaled(f[0]);
			static_for<1, Q, 2>([&](auto I) {
				constexpr int i = I;
				r[i] = S::encode_scaled(f[i+1]);
				r[i+1] = S::encode_scaled(f[i]);
			});
		}
	}
	if (FIELDS_ONLY || feat(F_UPDATE_FIELDS)) { // kernel.cpp:1581-1591, 1881-1893; a cell that does not write keeps its values
		const uint32_t x4 = 4u*x;
		if (!write_fields) { rhon = ld<float>(a.rho+row0, x4); uxn = ld<float>(a.u+row0, x4); uyn = ld<float>(a.u+a.N+row0, x4); uzn = ld<float>(a.u+2ull*a.N+row0, x4); }
		st(a.rho+row0, x4, rhon); st(a.u+row0, x4, uxn); st(a.u+a.N+row0, x4, uyn); st(a.u+2ull*a.N+row0, x4, uzn);
	}
	if constexpr (FIELDS_ONLY) return;

	// ---- streaming part 1: scatter to the slots that were read (kernel.cpp:1339-1345)
	const uint32_t sxb = fresh(xb), sxpb = fresh(xpb), sxmb = fresh(xmb);
	if constexpr (MEM == MEM_STORE_THROUGH && sizeof(raw) == 4) {
		// Written through at agent scope, every store instruction of a lane is a write of its own on the fabric: 4 bytes per lane
		// cost several times what 16 bytes per lane cost per byte.  The streams a wave stores at its own columns (14 of 19 for
		// D3Q19) are 256 contiguous, aligned bytes per wave: lanes 0..15 collect four neighbouring lanes' values each
		// (ds_bpermute) and store them as 16 bytes.  Waves with lanes past the row end keep the 4-byte stores.
		const uint32_t lane = threadIdx.x&63u;
		if (__ballot(true) == ~0ull && ((a.Nx|(uint32_t)a.S)&3u) == 0u) { // rows and streams that start on 16-byte boundaries
			const uint32_t from = (lane&15u)*16u, wave_xb = fresh((x-lane)*RB+lane*16u);
			typedef uint32_t quad __attribute__((ext_vector_type(4)));
			auto store16 = [&](raw* const base, const raw value) {
				const int v = (int)__builtin_bit_cast(uint32_t, value);
				quad q;
				q.x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, v); q.y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from+4u), v);
				q.z = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from+8u), v); q.w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from+12u), v);
				const uint64_t b = reinterpret_cast<uint64_t>(base); // wave-uniform; read into scalar registers for the store's base operand
				const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b>>32)), b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b); // the builtin returns int: no sign extension into the upper half
				const uint64_t sb = ((uint64_t)b_hi<<32)|(uint64_t)b_lo;
				if (lane < 16u) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(wave_xb), "v"(q), "s"(sb) : "memory"); // the wait state a 16-byte store needs before its data registers are written again
			};
			store16(pa[0], r[0]);
			static_for<1, Q, 2>([&](auto I) {
				constexpr int i = I, cx = dir<Q>(i,0);
				store16(pa[i], r[i]);
				if constexpr (cx == 0) store16(pa[i+1], r[i+1]);
				else st_ddf<MEM>(pa[i+1], cx > 0 ? sxpb : sxmb, r[i+1]); // one column to the side: 4 bytes per lane.  Measured and slower (79.8 % of 8 TB/s as it is): its 15 aligned groups of four as 16 bytes + four single cells 78.3 %; aligned pairs as 8 bytes from every second lane + two single cells 76.3 %
			});
			return;
		}
	}
	st_ddf<MEM>(pa[0], sxb, r[0]);
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0);
		st_ddf<MEM>(pa[i], sxb, r[i]);
		st_ddf<MEM>(pa[i+1], cx == 0 ? sxb : cx > 0 ? sxpb : sxmb, r[i+1]);
	});
}
template <int Q, int ST, int COLL, int FEAT, bool FIELDS_ONLY>
__global__ __launch_bounds__(256) void k_collide(const Args a) {
	// grid = (xchunks*Ny, planes): a launch may not exceed 2^32 work-items per grid dimension, and 1600^3 lattices do
	collide_row<Q, ST, COLL, FEAT, FIELDS_ONLY>(a, blockIdx.x%a.xchunks, blockIdx.x/a.xchunks, a.z0+blockIdx.y);
}

// elements v, v+1 of a V-element raw vector as one pair (16-bit formats: one dword; FP32: two floats)
template <class raw2, int V, class rvec> __device__ __forceinline__ raw2 pair_of(const rvec& d, const int v) {
	if constexpr (sizeof(raw2) == 4) { // 16-bit elements
		if constexpr (V == 2) return __builtin_bit_cast(uint32_t, d);
		else { typedef uint32_t wvec __attribute__((ext_vector_type(V/2))); return __builtin_bit_cast(wvec, d)[v/2]; }
	} else { raw2 r; r.x = d[v]; r.y = d[v+1]; return r; }
}
template <class raw2, int V, class rvec> __device__ __forceinline__ void set_pair(rvec& d, const int v, const raw2 p) {
	if constexpr (sizeof(raw2) == 4) {
		if constexpr (V == 2) d = __builtin_bit_cast(rvec, p);
		else { typedef uint32_t wvec __attribute__((ext_vector_type(V/2))); wvec w = __builtin_bit_cast(wvec, d); w[v/2] = p; d = __builtin_bit_cast(rvec, w); }
	} else { d[v] = p.x; d[v+1] = p.y; }
}

// ---------------------------------------------------------------------------------------------- stream_collide, V cells per lane, aligned vectors only
// Same result as k_collide<...,false>, arranged for the memory system: a lane owns V consecutive cells and every stream
// moves as aligned V*sizeof(raw)-byte vectors (4 cells of 16-bit DDFs: 512 contiguous bytes per wave instruction).  A slot stream that belongs
// to the +x (-x) neighbour is the same row shifted by one cell: the lane takes its own aligned vector and the first
// (last) element of the next (previous) lane's vector through a wave shuffle; only the lane at the end of a wave or of
// the row fetches that one element itself.  Stores mirror this.  The V cells are decoded, collided and encoded one after
// the other, the encoded result replacing the cell's element in the raw vectors (the register image of the in-place
// update in memory), which keeps the register count near that of one cell.  A skipped cell (solid, halo) keeps the bits
// that were read and they are written back unchanged: every slot has exactly one owner cell per step (kernel.cpp:1332-1345),
// so nothing else touches them in between.  Waves without any active cell leave before loading anything.
// P = 1: the V cells are collided one after the other; P = 2: two at a time in packed FP32 arithmetic (half the vector
// arithmetic instructions, twice the live registers) - same bits either way.
// FO = true is update_fields (kernel.cpp:1812-1894) with the same loads: moments only, rho/u written as whole vectors, no DDF stores (P = 1).
// collide_row_wide is the work of one workgroup (x-chunk `chunk` of lattice row (y, z)); MEM as for collide_row (the
// agent-scope forms move at most 4 bytes: two cells of 16-bit DDFs per lane).
template <int Q, int ST, int COLL, int FEAT, int V, int P, bool FO = false, int MEM = MEM_PLAIN>
__device__ __forceinline__ void collide_row_wide(const Args& a, const uint32_t chunk, const uint32_t y, const uint32_t z) {
	using S = Store<ST>;
	using raw = typename S::raw;
	static_assert(V == 2 || V == 4, "vectors of 2 or 4 cells");
	static_assert(P == 1 || P == 2, "cells collided together");
	static_assert(!FO || P == 1, "update_fields handles its cells one after the other");
	typedef raw rvec __attribute__((ext_vector_type(V)));
	typedef uint8_t fvec __attribute__((ext_vector_type(V)));
	typedef typename std::conditional<sizeof(raw) == 2, uint32_t, raw>::type lane_t; // what travels through a wave shuffle
	if (a.hy && (y == 0u || y == a.Ny-1u)) return;
	const uint32_t x = (chunk*blockDim.x+threadIdx.x)*V, lane = threadIdx.x&63u;
	const bool inrow = x < a.Nx; // lanes past the row end stay for the shuffles but touch no memory
	const uint32_t yq[3] = {y, y+1u == a.Ny ? 0u : y+1u, y == 0u ? a.Ny-1u : y-1u}; // index 0: same, 1: +1, 2: -1 (kernel.cpp:914-925)
	const uint32_t zq[3] = {z, z+1u == a.Nz ? 0u : z+1u, z == 0u ? a.Nz-1u : z-1u};
	const uint64_t row0 = ((uint64_t)z*a.Ny+y)*a.Nx;
	raw* const fi = static_cast<raw*>(a.fi);
	auto feat = [&]() { if constexpr (FEAT >= 0) return StaticFeat<(uint32_t)(FEAT >= 0 ? FEAT : 0)>{}; else return DynamicFeat{a.features}; }();

	fvec fl;
#pragma unroll
	for (int v = 0; v < V; v++) fl[v] = TYPE_S;
	if (inrow) fl = ld<fvec>(a.flags+row0, x);
	bool active[V];
	bool any = false;
#pragma unroll
	for (int v = 0; v < V; v++) {
		const uint32_t xv = x+(uint32_t)v;
		active[v] = inrow && (fl[v]&TYPE_BO) != TYPE_S && !(a.hx && (xv == 0u || xv == a.Nx-1u)); // kernel.cpp:1482-1485
		any = any || active[v];
	}
	if (__ballot(any) == 0ull) return;

	// the lane whose +x (-x) partner lane is in another wave or past the row end fetches / stores the seam element itself
	const bool seam_hi = lane == 63u || x+(uint32_t)V >= a.Nx, seam_lo = lane == 0u;
	const uint32_t x_hi = x+(uint32_t)V >= a.Nx ? x+(uint32_t)V-a.Nx : x+(uint32_t)V; // cell x+V, periodic
	const uint32_t x_lo = x == 0u ? a.Nx-1u : x-1u;                                       // cell x-1, periodic

	// ---- streaming part 2 (kernel.cpp:1332-1338): d[i] = raw DDFs of direction i for the V cells
	constexpr uint32_t RB = sizeof(raw);
	const uint32_t xb = x*RB, xhib = x_hi*RB, xlob = x_lo*RB; // byte offsets inside a row
	rvec d[Q];
	raw seam[Q];
	static_for<0, Q, 1>([&](auto I) { seam[I] = 0; });
	// wave-uniform row bases, formed once before the first load (with the scalar arithmetic out of the way loads and stores
	// issue back to back): pa[0] and pa[i] the lanes' own row in the streams they own, pa[i+1] the row that holds the +c_i
	// neighbours in the slot stream of pair (i, i+1)
	raw* pa[Q];
	pa[0] = fi+row0;
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		const uint32_t sA = a.odd ? i : i+1, sB = a.odd ? i+1 : i;
		pa[i] = fi+(uint64_t)sA*a.S+row0;
		pa[i+1] = fi+(uint64_t)sB*a.S+((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx;
	});
	auto own_row = [&](auto I) -> raw* { return pa[I]; };
	auto neighbour_row = [&](auto I) -> raw* { return pa[I+1]; };
	// Every group of accesses sits in one basic block together with its own copy of the lane offset (fresh): the
	// instruction selector then forms each address from a scalar base and that 32-bit offset, with no vector arithmetic.
	{ // lanes past the row end load the row's first vector (and use nothing of it): no registers to clear, no branch around the loads
		const uint32_t o = fresh(inrow ? xb : 0u);
		d[0] = ld_ddf<rvec, MEM>(pa[0], o);
		static_for<1, Q, 2>([&](auto I) {
			d[I] = ld_ddf<rvec, MEM>(own_row(I), o);
			d[I+1] = ld_ddf<rvec, MEM>(neighbour_row(I), o); // cells x..x+V-1 of the neighbour row; shifted below when cx != 0
		});
	}
	if (inrow) {
		if (seam_hi) {
			const uint32_t o = fresh(xhib);
			static_for<1, Q, 2>([&](auto I) { if constexpr (dir<Q>(I,0) > 0) seam[I+1] = ld_ddf<raw, MEM>(neighbour_row(I), o); });
		}
		if (seam_lo) {
			const uint32_t o = fresh(xlob);
			static_for<1, Q, 2>([&](auto I) { if constexpr (dir<Q>(I,0) < 0) seam[I+1] = ld_ddf<raw, MEM>(neighbour_row(I), o); });
		}
	}
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0);
		if constexpr (cx > 0) { // wanted: cells x+1..x+V
			const lane_t next_first = __shfl_down((lane_t)d[i+1][0], 1);
			const raw last = seam_hi ? seam[i+1] : (raw)next_first;
			rvec t;
#pragma unroll
			for (int v = 0; v+1 < V; v++) t[v] = d[i+1][v+1];
			t[V-1] = last;
			d[i+1] = t;
		} else if constexpr (cx < 0) { // wanted: cells x-1..x+V-2
			const lane_t prev_last = __shfl_up((lane_t)d[i+1][V-1], 1);
			const raw first = seam_lo ? seam[i+1] : (raw)prev_last;
			rvec t;
			t[0] = first;
#pragma unroll
			for (int v = 1; v < V; v++) t[v] = d[i+1][v-1];
			d[i+1] = t;
		}
	});

	// ---- the V cells, one after the other
	typedef float ovec __attribute__((ext_vector_type(V)));
	ovec out[4] = {ovec(0.0f), ovec(0.0f), ovec(0.0f), ovec(0.0f)}; // rho, ux, uy, uz of the V cells (UPDATE_FIELDS)
	bool wf[V];
#pragma unroll
	for (int v = 0; v < V; v++) wf[v] = false;
#pragma unroll
	for (int v = 0; v < V; v += P) {
		if constexpr (P == 1) {
			if (active[v]) {
				const uint32_t xv = x+(uint32_t)v;
				const uint8_t bo = fl[v]&TYPE_BO;
				float f[Q];
				constexpr bool halves = sizeof(raw) == 2; // the two 16-bit DDFs of a dword are decoded from it and encoded into it in place
				if constexpr (halves) static_for<0, Q, 1>([&](auto I) { f[I] = S::decode_half(pair_of<uint32_t, V>(d[I], v&~1), v&1); });
				else static_for<0, Q, 1>([&](auto I) { f[I] = S::decode(d[I][v]); });
				if (feat(F_MOVING_BOUNDARIES) && bo == TYPE_MS) apply_moving_boundaries<Q>(f, a, xv, yq, zq);
				float rhon, uxn, uyn, uzn;
				collide<Q, COLL, FO>(f, feat, &bo, a, row0, xv, rhon, uxn, uyn, uzn, &wf[v]);
				if (FO || feat(F_UPDATE_FIELDS)) { out[0][v] = rhon; out[1][v] = uxn; out[2][v] = uyn; out[3][v] = uzn; }
				if constexpr (!FO) { // streaming part 1 in registers: the slot that held f[i] receives f[i+1] and vice versa (kernel.cpp:1339-1345)
					S::template prescale<Q>(f);
					if constexpr (halves) {
						auto put = [&](rvec& into, const float value) { uint32_t w = pair_of<uint32_t, V>(into, v&~1); S::encode_half(w, v&1, value); set_pair<uint32_t, V>(into, v&~1, w); };
						put(d[0], f[0]);
						static_for<1, Q, 2>([&](auto I) { constexpr int i = I; put(d[i], f[i+1]); put(d[i+1], f[i]); });
					} else {
						d[0][v] = S::encode_scaled(f[0]);
						static_for<1, Q, 2>([&](auto I) {
							constexpr int i = I;
							d[i][v] = S::encode_scaled(f[i+1]);
							d[i+1][v] = S::encode_scaled(f[i]);
						});
					}
				}
			}
		} else {
			if (active[v] || active[v+1]) { // an inactive partner is computed along and its result dropped
				const uint32_t xv = x+(uint32_t)v;
				const uint8_t bo[2] = {(uint8_t)(fl[v]&TYPE_BO), (uint8_t)(fl[v+1]&TYPE_BO)};
				float2v f[Q];
				static_for<0, Q, 1>([&](auto I) { f[I] = S::decode2(pair_of<typename S::raw2, V>(d[I], v)); });
				if (feat(F_MOVING_BOUNDARIES) && (bo[0] == TYPE_MS || bo[1] == TYPE_MS)) {
#pragma unroll
					for (int k = 0; k < 2; k++) if (bo[k] == TYPE_MS) {
						float fk[Q];
						static_for<0, Q, 1>([&](auto I) { fk[I] = get_(f[I], k); });
						apply_moving_boundaries<Q>(fk, a, xv+(uint32_t)k, yq, zq);
						static_for<0, Q, 1>([&](auto I) { set_(f[I], k, fk[I]); });
					}
				}
				float2v rhon, uxn, uyn, uzn;
				collide<Q, COLL, false>(f, feat, bo, a, row0, xv, rhon, uxn, uyn, uzn, &wf[v]);
				wf[v] = wf[v] && active[v]; wf[v+1] = wf[v+1] && active[v+1];
				if (feat(F_UPDATE_FIELDS)) { // kernel.cpp:1581-1591; written pair by pair, which keeps four vectors of registers free
					float* const dst[4] = {a.rho+row0, a.u+row0, a.u+a.N+row0, a.u+2ull*a.N+row0};
					float2v val[4] = {rhon, uxn, uyn, uzn};
					if (!(wf[v] && wf[v+1])) { // a cell that does not write keeps its value; the pair is still written as a whole (see below)
#pragma unroll
						for (int k = 0; k < 4; k++) {
							const float2v old = ld<float2v>(dst[k], 4u*xv);
							if (!wf[v]) val[k].x = old.x;
							if (!wf[v+1]) val[k].y = old.y;
						}
					}
#pragma unroll
					for (int k = 0; k < 4; k++) st(dst[k], 4u*xv, val[k]);
				}
				// streaming part 1 in registers (kernel.cpp:1339-1345); an inactive cell keeps the bits that were read
				const bool keep0 = !active[v], keep1 = !active[v+1];
				S::template prescale<Q>(f);
				auto put = [&](auto I, const float2v val) {
					const typename S::raw2 e = S::encode2_scaled(val);
					set_pair<typename S::raw2, V>(d[I], v, S::merge2(pair_of<typename S::raw2, V>(d[I], v), e, keep0, keep1));
				};
				put(std::integral_constant<int, 0>{}, f[0]);
				static_for<1, Q, 2>([&](auto I) {
					constexpr int i = I;
					put(std::integral_constant<int, i>{}, f[i+1]);
					put(std::integral_constant<int, i+1>{}, f[i]);
				});
			} else if (feat(F_UPDATE_FIELDS) && inrow) { // a pair without active cell: its rho/u are re-written unchanged so that memory sectors are written whole
				const uint32_t o4 = 4u*(x+(uint32_t)v);
				float* const dst[4] = {a.rho+row0, a.u+row0, a.u+a.N+row0, a.u+2ull*a.N+row0};
#pragma unroll
				for (int k = 0; k < 4; k++) { const float2v old = ld<float2v>(dst[k], o4); st(dst[k], o4, old); }
			}
		}
		__builtin_amdgcn_sched_barrier(0); // keep the cells apart: interleaving them multiplies the live registers
	}

	// ---- rho/u output (kernel.cpp:1581-1591): always whole aligned vectors.  Cells that do not write (solid, halo,
	// equilibrium boundary) get the value they already hold: partially written memory sectors next to obstacles cost
	// far more than re-writing a few unchanged values (UPDATE_FIELDS, FP16S, 6 % solid cells: 47 400 -> 55 100 MLUPs/s; without solids 56 100)
	if (P == 1 && (FO || feat(F_UPDATE_FIELDS)) && inrow) {
		bool all = true;
#pragma unroll
		for (int v = 0; v < V; v++) all = all && wf[v];
		float* const dst[4] = {a.rho+row0, a.u+row0, a.u+a.N+row0, a.u+2ull*a.N+row0};
		if (!all) {
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const ovec old = ld<ovec>(dst[k], 4u*x);
#pragma unroll
				for (int v = 0; v < V; v++) if (!wf[v]) out[k][v] = old[v];
			}
		}
#pragma unroll
		for (int k = 0; k < 4; k++) st(dst[k], 4u*x, out[k]);
	}

	if constexpr (FO) return;

	// ---- streaming part 1 (kernel.cpp:1339-1345): write every vector back to where it came from
	lane_t carry[Q]; // for shifted streams: the element that belongs into the neighbouring lane's aligned vector
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0);
		carry[i+1] = 0;
		if constexpr (cx > 0) carry[i+1] = __shfl_up((lane_t)d[i+1][V-1], 1); // previous lane's value for cell x
		if constexpr (cx < 0) carry[i+1] = __shfl_down((lane_t)d[i+1][0], 1); // next lane's value for cell x+V-1
	});
	if (!inrow) return;
	bool packed = false;
	if constexpr (MEM == MEM_STORE_THROUGH && sizeof(rvec) == 4 && ST == ST_FP16S) {
		// written through at agent scope, 4 bytes per lane are a write of their own on the fabric each: full waves store the
		// aligned vectors of four lanes together, 16 bytes from each of the lanes 0..15 (see collide_row).  512^3 FP16S:
		// 74 900 -> 75 500 MLUPs/s; FP16C, whose codec already fills the vector issue slots, loses 1 % and keeps 4-byte stores
		if (__ballot(true) == ~0ull) {
			packed = true;
			const uint32_t from = (lane&15u)*16u, wave_xb = fresh((x-lane*(uint32_t)V)*RB+lane*16u);
			typedef uint32_t quad __attribute__((ext_vector_type(4)));
			auto store16 = [&](raw* const base, const rvec value) {
				const int v = (int)__builtin_bit_cast(uint32_t, value);
				quad q;
				q.x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, v); q.y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from+4u), v);
				q.z = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from+8u), v); q.w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from+12u), v);
				const uint64_t b = reinterpret_cast<uint64_t>(base);
				const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b>>32)), b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
				const uint64_t sb = ((uint64_t)b_hi<<32)|(uint64_t)b_lo;
				if (lane < 16u) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(wave_xb), "v"(q), "s"(sb) : "memory");
			};
			store16(pa[0], d[0]);
			static_for<1, Q, 2>([&](auto I) {
				store16(own_row(I), d[I]);
				if constexpr (dir<Q>(I,0) == 0) store16(neighbour_row(I), d[I+1]);
			});
		}
	}
	if (!packed) { // own slots and the neighbour slots of directions without x component: whole aligned vectors
		const uint32_t o = fresh(xb);
		st_ddf<MEM>(pa[0], o, d[0]);
		static_for<1, Q, 2>([&](auto I) {
			st_ddf<MEM>(own_row(I), o, d[I]);
			if constexpr (dir<Q>(I,0) == 0) st_ddf<MEM>(neighbour_row(I), o, d[I+1]);
		});
	}
	// directions with c_x > 0: d[i+1] holds cells x+1..x+V
	if (!seam_lo) {
		const uint32_t o = fresh(xb);
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I;
			if constexpr (dir<Q>(i,0) > 0) {
				rvec t;
				t[0] = (raw)carry[i+1];
#pragma unroll
				for (int v = 1; v < V; v++) t[v] = d[i+1][v-1];
				st_ddf<MEM>(neighbour_row(I), o, t);
			}
		});
	} else { // cell x comes from the previous wave's last lane
		const uint32_t o = fresh(xb);
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I;
			if constexpr (dir<Q>(i,0) > 0) {
#pragma unroll
				for (int v = 1; v < V; v++) st_ddf<MEM>(neighbour_row(I), o+(uint32_t)v*RB, (raw)d[i+1][v-1]);
			}
		});
	}
	if (seam_hi) {
		const uint32_t o = fresh(xhib);
		static_for<1, Q, 2>([&](auto I) { constexpr int i = I; if constexpr (dir<Q>(i,0) > 0) st_ddf<MEM>(neighbour_row(I), o, (raw)d[i+1][V-1]); });
	}
	// directions with c_x < 0: d[i+1] holds cells x-1..x+V-2
	if (!seam_hi) {
		const uint32_t o = fresh(xb);
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I;
			if constexpr (dir<Q>(i,0) < 0) {
				rvec t;
#pragma unroll
				for (int v = 0; v+1 < V; v++) t[v] = d[i+1][v+1];
				t[V-1] = (raw)carry[i+1];
				st_ddf<MEM>(neighbour_row(I), o, t);
			}
		});
	} else { // cell x+V-1 comes from the next wave's first lane
		const uint32_t o = fresh(xb);
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I;
			if constexpr (dir<Q>(i,0) < 0) {
#pragma unroll
				for (int v = 0; v+1 < V; v++) st_ddf<MEM>(neighbour_row(I), o+(uint32_t)v*RB, (raw)d[i+1][v+1]);
			}
		});
	}
	if (seam_lo) {
		const uint32_t o = fresh(xlob);
		static_for<1, Q, 2>([&](auto I) { constexpr int i = I; if constexpr (dir<Q>(i,0) < 0) st_ddf<MEM>(neighbour_row(I), o, (raw)d[i+1][0]); });
	}
}

template <int Q, int ST, int COLL, int FEAT, int V, int P, bool FO = false>
__global__ __launch_bounds__(256, (P == 2 && V == 2 ? 3 : 1)) void k_collide_wide(const Args a) { // two cells in packed arithmetic: at least 3 waves per SIMD (at most 168 registers; 4 cells per lane would spill)
	collide_row_wide<Q, ST, COLL, FEAT, V, P, FO>(a, blockIdx.x%a.xchunks, blockIdx.x/a.xchunks, a.z0+blockIdx.y);
}

// ---------------------------------------------------------------------------------------------- two time steps in one launch
// k_collide_pair runs step t and step t+1 in ONE launch - of a whole lattice that is periodic in z, or of the inner
// planes of a z-slab between two halo exchanges - step t+1 trailing step t by a few planes, so that step t+1 finds the
// DDFs step t wrote in the 256 MiB Infinity Cache instead of in HBM (the reference launches one kernel per step,
// lbm.cpp:1411-1459; the result is the same bits, since every cell still sees exactly the values the previous step left).
//   Order of work (blockIdx.y = plane task, dispatched in ascending order), for the whole periodic lattice: step t on
//   planes 0..lag+1, then alternately step t+1 on plane p and step t on plane p+lag, at the end step t+1 on the planes
//   next to the periodic seam (Nz-lag .. Nz-1, 0, 1).  PairArgs describes the general form (two plane ranges).  A
//   workgroup works on one row chunk of one step: with both steps' rows in one workgroup the second step's loads would
//   queue behind the drain of the first step's stores (measured: 67 % of 8 TB/s against 78 % with nothing waiting at all).
//   Dependence: a cell of step t+1 in plane p touches slots in planes p-1..p+1, whose step-t owners sit in planes p-2..p+2
//   (esoteric-pull: a slot belongs to the cell at its place or to the cell one step against c_i, kernel.cpp:1332-1345).
//   Every step-t workgroup counts itself done (a counter per plane and shard; the last arriver of a shard reports to the
//   plane's word), every step-t+1 workgroup waits until the five planes p-2..p+2 have all their shards reported.  With
//   lag >= 3 a workgroup only ever waits for workgroups with smaller grid indices, which have been dispatched before it.
//   The wait is bounded: when it runs out the workgroup raises *error, writes its task to the redo list and computes nothing;
//   every other wait that finds a plane unfinished looks at *error and gives up too.  k_collide_seam, launched behind, runs
//   the tasks of the list (step t is complete by then, nothing in the launch depends on step t+1): the lattice is right in
//   every case, only time is lost, and the host turns the pairing off when it finds *error set.
//   Visibility between workgroups of the launch (per-XCD L2s are not coherent, a CU's L1 is never refreshed): step t
//   stores its DDFs written through at agent scope, drains them (s_waitcnt vmcnt(0) in every wave, then a workgroup
//   barrier) before one lane counts the workgroup done with an agent-scope atomic; step t+1 polls with agent-scope loads
//   (wave 0, with growing pauses: most workgroups of an XCD that is ahead find the XCD that is behind unfinished),
//   passes a workgroup barrier and loads its DDFs at agent scope.  Step t+1 stores plainly: the next launch reads that.
//   Step t never reads a byte written in this launch (it runs ahead of step t+1 by more than the two planes step t+1
//   writes beyond its own), so its loads are plain.
// Not for UPDATE_FIELDS (both steps would write rho/u of a cell from different XCDs with no order between the write-backs).
struct PairArgs {
	uint32_t* plane_done; // PAIR_REPLICAS arrays, replica_stride words apart, of Nz words: shards of the plane that have reported (zeroed before the launch)
	uint32_t replica_stride;
	uint32_t* shard_done; // Nz*PAIR_SHARDS counters, 64 bytes apart (zeroed before the launch)
	uint32_t* error;      // raised by a wait that ran out (zeroed before the launch)
	uint32_t* redo_count; // tasks of step t+1 given up (zeroed before the launch)
	uint64_t* redo;       // their (z*Ny+y)*xchunks+chunk, for k_collide_seam
	uint32_t lag;         // >= 3
	uint32_t spin_limit;  // polls before a wait gives up
	// Step t runs on the planes first1 .. first1+count1-1, step t+1 on first2 .. first2+count2-1 (taken modulo Nz).  The whole
	// periodic lattice: first1 = 0, count1 = Nz, first2 = 2, count2 = Nz (planes 0 and 1 come last).  The inner planes of a
	// z-slab whose planes next to the halos are stepped by launches of their own: first2 = first1+2, count2 = count1-4, so
	// that every plane a step-t+1 cell depends on (p-2 .. p+2) belongs to this launch's step t.
	uint32_t first1, count1, first2, count2;
};
// The plane words are kept in PAIR_REPLICAS copies on cache lines of their own: every waiting workgroup of a plane asks
// about the same five words, and a thousand workgroups asking one line queue up at its memory channel.
constexpr uint32_t PAIR_SHARDS = 8u, PAIR_SHARD_STRIDE = 16u, PAIR_REPLICAS = 16u;
// V = 1: the rows of k_collide; V = 2 (16-bit DDFs): the rows of k_collide_wide, P cells collided together.
template <int Q, int ST, int COLL, int FEAT, int V = 1, int P = 1>
__global__ __launch_bounds__(256, (P == 2 && V == 2 ? 3 : 1)) void k_collide_pair(const Args a, const PairArgs p) {
	__shared__ uint32_t give_up;
	const uint32_t Nz = a.Nz, task = blockIdx.y, per_plane = gridDim.x, in_plane = blockIdx.x;
	// step t+1 on plane first2+j follows step t on plane first2+j+lag-1: `ahead` planes of step t come first, then the two
	// alternate, then whichever step still has planes left finishes (gridDim.y = count1+count2)
	const uint32_t ahead = min(p.count1, p.first2+p.lag-p.first1), both = min(p.count1-ahead, p.count2);
	uint32_t z, second = 0u;
	if (task < ahead) z = p.first1+task;
	else if (task < ahead+2u*both) { const uint32_t k = task-ahead; second = (k&1u)^1u; z = second ? p.first2+k/2u : p.first1+ahead+k/2u; }
	else if (ahead+both < p.count1) z = p.first1+ahead+both+(task-ahead-2u*both); // step t+1 has run out of planes first
	else { second = 1u; z = p.first2+both+(task-ahead-2u*both); }
	if (z >= Nz) z -= Nz;
	const uint32_t chunk = in_plane%a.xchunks, y = in_plane/a.xchunks;
	if (!second) {
		__builtin_amdgcn_s_setprio(2); // the first step is what everything else waits for: its waves issue ahead of the second step's (512^3 FP32: 72.5 -> 77.8 % of 8 TB/s)
		if constexpr (V == 1) collide_row<Q, ST, COLL, FEAT, false, MEM_STORE_THROUGH>(a, chunk, y, z);
		else collide_row_wide<Q, ST, COLL, FEAT, V, P, false, MEM_STORE_THROUGH>(a, chunk, y, z);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores have been acknowledged
		__syncthreads();
		if (threadIdx.x < 64u) {
			const uint32_t shard = in_plane%PAIR_SHARDS, members = per_plane/PAIR_SHARDS+(shard < per_plane%PAIR_SHARDS ? 1u : 0u);
			uint32_t before = 0u;
			if (threadIdx.x == 0u) before = __hip_atomic_fetch_add(p.shard_done+((uint64_t)z*PAIR_SHARDS+shard)*PAIR_SHARD_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const bool last = __shfl((int)(before+1u == members), 0) != 0; // the shard's last arriver reports to every replica of the plane's word
			if (last && threadIdx.x < PAIR_REPLICAS) __hip_atomic_fetch_add(p.plane_done+(uint64_t)threadIdx.x*p.replica_stride+z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	} else {
		// the row's flags (never written by a step) are fetched while the question below is under way
		const uint32_t x = (chunk*blockDim.x+threadIdx.x)*(uint32_t)V;
		uint32_t early = 0u;
		if (x < a.Nx) early = ld<uint8_t>(a.flags+((uint64_t)z*a.Ny+y)*a.Nx, x);
		if (threadIdx.x < 64u) {
			const uint32_t lane = threadIdx.x, shards = per_plane < PAIR_SHARDS ? per_plane : PAIR_SHARDS; // shards without members never report
			const uint32_t zz = (z+Nz-2u+(lane < 5u ? lane : 0u))%Nz;
			const bool asked = lane < 5u && (zz+Nz-p.first1)%Nz < p.count1; // planes outside this launch's step t were finished by earlier launches
			const uint32_t* const word = lane == 5u ? p.error : p.plane_done+(uint64_t)(in_plane%PAIR_REPLICAS)*p.replica_stride+zz;
			bool failed = false;
			for (uint32_t spins = 0u; ; spins++) {
				uint32_t v = 0u;
				if (asked || (lane == 5u && spins > 0u)) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the error word only once a plane was found unfinished
				if (__ballot(asked && v < shards) == 0ull) break;
				if (__ballot(lane == 5u && v != 0u) != 0ull || spins >= p.spin_limit) {
					if (lane == 0u) __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					failed = true;
					break;
				}
				if (spins < 2u) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127); // a workgroup that is ahead of the first step asks rarely: its questions share the memory system with the work it waits for
			}
			if (lane == 0u) {
				give_up = failed ? 1u : 0u;
				if (failed) p.redo[__hip_atomic_fetch_add(p.redo_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = (((uint64_t)a.Nz+z)*a.Ny+y)*a.xchunks+chunk; // a task of step t+1
			}
		}
		__syncthreads();
		if (give_up) return;
		Args b = a;
		b.odd = a.odd^1u;
		if constexpr (V == 1) collide_row<Q, ST, COLL, FEAT, false, MEM_LOAD_COHERENT, true>(b, chunk, y, z, early);
		else {
			asm volatile("" :: "v"(early)); // the early load has brought the lanes' flag bytes into the cache
			collide_row_wide<Q, ST, COLL, FEAT, V, P, false, MEM_LOAD_COHERENT>(b, chunk, y, z);
		}
	}
}

// ---------------------------------------------------------------------------------------------- two (to four) time steps in one launch, band by band on each XCD
// k_collide_band is the second form of two steps per launch: step t+1 finds what step t wrote in the L2 of its own XCD (4 MiB)
// instead of in the Infinity Cache, and what it writes over those lines leaves the L2 once instead of twice.
//   Which cells depend on which: between two consecutive steps the owner of a DDF slot moves by exactly one c_i (stream i at
//   place m belongs to cell m at one parity of t and to cell m-c_i at the other, kernel.cpp:1332-1345), so a cell of step t+1
//   at (x, y, z) needs step t finished on the rows y-1..y+1 of the planes z-1..z+1 and on nothing else.
//   Order of work: the rows of the lattice are cut into eight blocks, one per XCD (workgroups are dealt to the XCDs round-robin
//   by their linear index, which blockIdx.x % 8 stands for here), a block into bands of R rows.  An XCD sweeps band after
//   band along z, step t+1 trailing step t by `lag` planes and shifted down by one row (its rows b0-1..b1-2 need step t on
//   b0-2..b1-1: the band itself and the two rows above the band swept before), so that the lines between the two steps are a
//   few hundred KB per plane of the band.  The first and the last row of every block need step t from a neighbouring XCD:
//   their step t+1 is left to k_collide_seam, a launch of its own behind this one (16 of Ny rows per plane).
//   Hand-over inside an XCD: step t stores plainly (the lines stay in the XCD's L2, dirty), every wave drains its stores
//   (s_waitcnt vmcnt(0): acknowledged by the L2), the workgroup passes a barrier and one lane stores the workgroup's stamp -
//   the launch's number and the XCD it ran on, read from the hardware register, not inferred - again plainly, into the word
//   of its task.  Step t+1 loads the stamps of the (at most 27) step-t tasks it depends on - rows y-1..y+1, planes z-1..z+1,
//   its own x chunk and the two beside it - past the L1 (sc1: served by the L2, the one copy all CUs of an XCD share; the same
//   order makes workgroup scope coherent when the waves of one workgroup run on several CUs of an XCD), passes a barrier and
//   loads its DDFs the same way.  Stamps and DDFs never leave the XCD between the two steps: no atomic, no word that the
//   whole chip polls (counters that 500 000 workgroups add to and ask about through one memory channel cost more than the
//   second step's reads save: 50 % of 8 TB/s against 86 % with the waits switched off).  A stamp of this launch that names
//   another XCD means the workgroups were not dealt as assumed; then, and when a wait runs out, the workgroup raises *error,
//   writes its task to the redo list and computes nothing, and every other wait that finds a stamp missing looks at *error
//   and gives up too.  k_collide_seam runs the tasks of the list (step t is complete by then and nothing in the launch depends
//   on step t+1), so the lattice is right in every case and only time is lost; the host turns the pairing off when it finds
//   *error set.
//   More than two steps (BandArgs::steps, up to 4) go the same way: step t+s takes the band's rows shifted down by s, trails
//   step t+s-1 by `lag` planes, waits for that step's stamps and leaves stamps of its own; s rows at either edge of a block are
//   left to the s-th k_collide_seam launch, and a task that gives up leaves no stamp, so the tasks that need it give up too.
//   512^3 FP32: 3 steps 85.9-87.8 %, 4 steps 82.9-85.3 % of 8 TB/s where 2 steps reach 88.3-89.3 % - the second step's
//   reads already come from a cache, further steps only add waits and seam rows - so the host asks for 2 unless told otherwise.
struct BandArgs {
	uint32_t* stamps;      // [steps-1][8 blocks][bands][Nz][R][xchunks]: (launch number << 3 | XCD) of every finished task of a step that another step of the launch follows; never cleared, launch numbers rise
	uint64_t stamps_per_step;
	uint32_t stamp;        // this launch's number << 3
	uint32_t* error;       // raised by a wait that ran out (zeroed before the launch)
	uint32_t* redo_count;  // tasks given up (zeroed before the launch)
	uint64_t* redo;        // their ((step*Nz+z)*Ny+y)*xchunks+chunk
	uint32_t block_rows;   // rows per XCD block
	uint32_t R, bands;     // rows per band, bands per block
	uint32_t lag;          // >= 2: step t+s on plane u is dispatched lag-1 planes behind step t+s-1 on plane u+1
	uint32_t spin_limit;   // polls before a wait gives up
	uint32_t steps;        // 2..4 time steps in the launch
	uint32_t first[4], count[4]; // planes of step t+s: first[s] .. first[s]+count[s]-1 modulo Nz, first[s] >= first[s-1]+1 (not reduced modulo Nz)
};
#ifndef LBM_BAND_EXP
#define LBM_BAND_EXP 0 // diagnostic builds (tools/band_variants.sh): 1 no waits, 2 plain loads in the later steps, 4 write-through stores in step t, 8 no drain and no stamp after a step, 32 wait statistics - 1 and 8 give wrong lattices
#endif
__device__ __forceinline__ uint32_t xcd_of_this_wave() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return x&7u; }
#ifndef LBM_BAND_BLOCKS
#define LBM_BAND_BLOCKS 1 // the same for the other instances (diagnostic builds)
#endif
#ifndef LBM_BAND_P2_BLOCKS
#define LBM_BAND_P2_BLOCKS 3 // workgroups per CU the register allocation of the two-cells-together instances aims at (diagnostic builds: 5 = 88 registers, 6 spills)
#endif
template <int Q, int ST, int COLL, int FEAT, int V = 1, int P = 1>
__global__ __launch_bounds__(256, (P == 2 && V == 2 ? LBM_BAND_P2_BLOCKS : LBM_BAND_BLOCKS)) void k_collide_band(const Args a, const BandArgs p) {
	__shared__ uint32_t give_up;
	const uint32_t Nz = a.Nz, block_of = blockIdx.x&7u, within = blockIdx.x>>3, band = blockIdx.z; // grid = (8*R*xchunks, plane times * steps, bands)
	const uint32_t bs = block_of*p.block_rows, be = min(bs+p.block_rows, a.Ny); // this block's rows
	const uint32_t b0 = bs+band*p.R, b1 = min(b0+p.R, be);                        // this band's rows
	if (bs >= a.Ny || b0 >= be) return;
	// plane time tau: step t+s works on plane first[0]+tau-s*lag, the latest step first
	const uint32_t tau = blockIdx.y/p.steps, s = p.steps-1u-blockIdx.y%p.steps;
	const int j = (int)tau-(int)(s*p.lag)-(int)(p.first[s]-p.first[0]);
	if (j < 0 || j >= (int)p.count[s]) return;
	const uint32_t z = (p.first[s]+(uint32_t)j)%Nz;
	const uint32_t chunk = within%a.xchunks, row = within/a.xchunks, xcd = xcd_of_this_wave();
	// step t+s takes the band's rows shifted down by s; s rows at either edge of the block are not this launch's
	if (b0+row >= b1 || b0+row < bs+2u*s) return; // bs+s <= y; y < be-s follows from b1 <= be
	const uint32_t y = b0+row-s;
	if (s == 0u) {
		__builtin_amdgcn_s_setprio(2); // step t is what the other steps wait for
		constexpr int MEM1 = (LBM_BAND_EXP&4) ? MEM_STORE_THROUGH : MEM_PLAIN;
		if constexpr (V == 1) collide_row<Q, ST, COLL, FEAT, false, MEM1>(a, chunk, y, z);
		else collide_row_wide<Q, ST, COLL, FEAT, V, P, false, MEM1>(a, chunk, y, z);
	} else {
		const uint32_t x = (chunk*blockDim.x+threadIdx.x)*(uint32_t)V;
		uint32_t early = 0u;
		if (x < a.Nx) early = ld<uint8_t>(a.flags+((uint64_t)z*a.Ny+y)*a.Nx, x); // flags are never written by a step
		if (threadIdx.x < 64u) {
			// lane = (plane z-1..z+1) * 9 + (row y-1..y+1) * 3 + (chunk-1..chunk+1, periodic; the same chunk twice in short rows is harmless)
			const uint32_t lane = threadIdx.x, dz = lane/9u, dr = (lane/3u)%3u, dc = lane%3u;
			const uint32_t zz = (z+Nz-1u+(lane < 27u ? dz : 0u))%Nz;
			const bool asked = lane < 27u && (zz+Nz-p.first[s-1u]%Nz)%Nz < p.count[s-1u]; // planes outside this launch's step t+s-1 were finished by earlier launches
			// step t+s-1 did row y' as row y'+s-1 of the block's bands; y-1 >= bs+s-1 and y+1 < be-s+1 lie inside what it did
			const uint32_t at = y-1u+(s-1u)-bs+dr, bb = at/p.R, rr = at%p.R;
			const uint32_t cc = (chunk+a.xchunks+dc-1u)%a.xchunks;
			const uint32_t* const word = lane == 27u ? p.error : p.stamps+(s-1u)*p.stamps_per_step+((((uint64_t)block_of*p.bands+bb)*Nz+zz)*p.R+rr)*a.xchunks+cc;
			const uint32_t want = p.stamp|xcd;
			bool failed = false;
#if LBM_BAND_EXP&32
			const uint64_t t_begin = __builtin_readcyclecounter();
			uint32_t polls = 0u;
#endif
			for (uint32_t spins = 0u; !(LBM_BAND_EXP&1); spins++) {
#if LBM_BAND_EXP&32
				polls++;
#endif
				uint32_t v = 0u;
				if (asked || (lane == 27u && spins > 0u)) v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the error word only once a stamp was found missing
				if (__ballot(asked && v != want) == 0ull) break;
				const bool elsewhere = asked && (v&~7u) == p.stamp && v != want; // finished, on another XCD: its DDFs may sit in that XCD's L2
				if (__ballot(elsewhere || (lane == 27u && v != 0u)) != 0ull || spins >= p.spin_limit) {
					if (lane == 0u) __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					failed = true;
					break;
				}
				if (spins < 2u) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127);
			}
#if LBM_BAND_EXP&32
			if (lane == 0u) { // header words 4..: waiting workgroups, polls, workgroups that polled more than once / more than three times, cycles between the first poll and the last
				__hip_atomic_fetch_add(p.error+4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_add(p.error+5, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (polls > 1u) __hip_atomic_fetch_add(p.error+6, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p.error+8), (unsigned long long)(__builtin_readcyclecounter()-t_begin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (polls > 3u) __hip_atomic_fetch_add(p.error+7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
#endif
			if (lane == 0u) {
				give_up = failed ? 1u : 0u;
				if (failed) p.redo[__hip_atomic_fetch_add(p.redo_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = (((uint64_t)s*Nz+z)*a.Ny+y)*a.xchunks+chunk;
			}
		}
		__syncthreads();
		if (give_up) return; // without a stamp: the tasks that depend on this one give up as well
		if (s+1u < p.steps) __builtin_amdgcn_s_setprio(2);
		Args b = a;
		b.odd = a.odd^(s&1u);
		constexpr int MEM2 = (LBM_BAND_EXP&2) ? MEM_PLAIN : MEM_LOAD_COHERENT;
		if constexpr (V == 1) collide_row<Q, ST, COLL, FEAT, false, MEM2, true>(b, chunk, y, z, early);
		else {
			asm volatile("" :: "v"(early));
			collide_row_wide<Q, ST, COLL, FEAT, V, P, false, MEM2>(b, chunk, y, z);
		}
	}
	if (s+1u < p.steps && !(LBM_BAND_EXP&8)) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores have been acknowledged by the L2
		__syncthreads();
		if (threadIdx.x == 0u) *(volatile uint32_t*)(p.stamps+s*p.stamps_per_step+((((uint64_t)block_of*p.bands+band)*Nz+z)*p.R+row)*a.xchunks+chunk) = p.stamp|xcd;
	}
}
// What k_collide_band leaves of step t+s: the rows at the edges of the XCD blocks (blockIdx.y < count: plane first+blockIdx.y,
// blockIdx.x = row of the list * xchunks + chunk), and that step's tasks on the redo list (blockIdx.y == count, the workgroups
// stride over the list).  `a` is step t+s's (odd set by the host).  Plain accesses: a launch of its own, one per step, in order.
struct SeamArgs {
	uint32_t rows[64], nrows;
	uint32_t first, count, step;
	const uint32_t* redo_count;
	const uint64_t* redo;
};
template <int Q, int ST, int COLL, int FEAT, int V = 1, int P = 1>
__global__ __launch_bounds__(256, (P == 2 && V == 2 ? 3 : 1)) void k_collide_seam(const Args a, const SeamArgs p) {
	if (blockIdx.y < p.count) {
		const uint32_t row = blockIdx.x/a.xchunks, chunk = blockIdx.x%a.xchunks;
		if (row >= p.nrows) return;
		const uint32_t z = (p.first+blockIdx.y)%a.Nz;
		if constexpr (V == 1) collide_row<Q, ST, COLL, FEAT, false>(a, chunk, p.rows[row], z);
		else collide_row_wide<Q, ST, COLL, FEAT, V, P>(a, chunk, p.rows[row], z);
	} else {
		const uint32_t n = *p.redo_count;
		for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
			const uint64_t e = p.redo[k];
			const uint32_t chunk = (uint32_t)(e%a.xchunks);
			const uint64_t zy = e/a.xchunks;
			const uint32_t y = (uint32_t)(zy%a.Ny), sz = (uint32_t)(zy/a.Ny);
			if (sz/a.Nz != p.step) continue;
			if constexpr (V == 1) collide_row<Q, ST, COLL, FEAT, false>(a, chunk, y, sz%a.Nz);
			else collide_row_wide<Q, ST, COLL, FEAT, V, P>(a, chunk, y, sz%a.Nz);
		}
	}
}

#ifdef LBM_SMALL_KERNELS // the one-cell-per-thread kernels are compiled once, in lbmhip.hip
// ---------------------------------------------------------------------------------------------- one-cell-per-thread helpers
struct Cell { uint32_t x, y, z; uint64_t n; bool valid; };
__device__ __forceinline__ Cell cell_of_thread(const Args& a) { // grid = (xchunks*Ny, Nz), like the step kernels: y and z are wave-uniform, no 64-bit division
	Cell c;
	const uint32_t chunk = blockIdx.x%a.xchunks;
	c.y = blockIdx.x/a.xchunks; c.z = blockIdx.y;
	c.x = chunk*blockDim.x+threadIdx.x;
	c.valid = c.x < a.Nx;
	c.n = ((uint64_t)c.z*a.Ny+c.y)*a.Nx+c.x;
	return c;
}
__device__ __forceinline__ bool is_halo(const Args& a, const Cell& c) { // kernel.cpp:850-853
	return (a.hx && (c.x == 0u || c.x >= a.Nx-1u)) || (a.hy && (c.y == 0u || c.y >= a.Ny-1u)) || (a.hz && (c.z == 0u || c.z >= a.Nz-1u));
}
template <int Q> __device__ __forceinline__ void neighbors(const Args& a, uint32_t x, uint32_t y, uint32_t z, uint64_t* j) { // kernel.cpp:926-968
	const uint32_t xq[3] = {x, x+1u == a.Nx ? 0u : x+1u, x == 0u ? a.Nx-1u : x-1u};
	const uint32_t yq[3] = {y, y+1u == a.Ny ? 0u : y+1u, y == 0u ? a.Ny-1u : y-1u};
	const uint32_t zq[3] = {z, z+1u == a.Nz ? 0u : z+1u, z == 0u ? a.Nz-1u : z-1u};
	static_for<0, Q, 1>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0), cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		j[i] = ((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx+xq[cx == 0 ? 0 : cx > 0 ? 1 : 2];
	});
}
template <int Q, int ST> __device__ __forceinline__ void load_f(const Args& a, uint64_t n, const uint64_t* j, uint32_t odd, float* f) {
	using S = Store<ST>; using raw = typename S::raw;
	const raw* fi = static_cast<const raw*>(a.fi);
	f[0] = S::decode(fi[n]);
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I;
		f[i]   = S::decode(fi[(uint64_t)(odd ? i : i+1)*a.S+n]);
		f[i+1] = S::decode(fi[(uint64_t)(odd ? i+1 : i)*a.S+j[i]]);
	});
}

// FluidX3D-master/src/kernel.cpp:1364-1446 (without SURFACE/TEMPERATURE).
// What a cell does before it stores its equilibrium DDFs: returns false for a cell the kernel skips (halo column), else its
// density and velocity (cleared for solid cells as the reference does).  The owner lane also writes the cleared velocity
// and the TYPE_MS mark back; a lane that only needs a neighbour's values (see k_initialize) passes owner = false.
template <int Q> __device__ __forceinline__ bool initial_state(const Args& a, const uint32_t x, const uint32_t y, const uint32_t z, const uint64_t row0, const bool owner,
                                                              float& rho, float& ux, float& uy, float& uz) {
	if (a.hx && (x == 0u || x >= a.Nx-1u)) return false; // kernel.cpp:850-853
	const uint32_t x4 = 4u*x;
	uint8_t fl = ld<uint8_t>(a.flags+row0, x);
	rho = ld<float>(a.rho+row0, x4);
	ux = ld<float>(a.u+row0, x4); uy = ld<float>(a.u+a.N+row0, x4); uz = ld<float>(a.u+2ull*a.N+row0, x4);
	const uint8_t bo = fl&TYPE_BO;
	const bool moving_ext = (a.features&F_MOVING_BOUNDARIES) != 0u;
	if (bo == TYPE_S) {
		bool only_solid = true;
		if (moving_ext) { // otherwise the velocity of every solid cell is cleared and the neighbours do not matter
			uint64_t j[Q]; neighbors<Q>(a, x, y, z, j);
			static_for<1, Q, 1>([&](auto I) { only_solid = only_solid && (a.flags[j[I]]&TYPE_BO) == TYPE_S; });
		}
		if (only_solid || !moving_ext) {
			ux = uy = uz = 0.0f;
			if (owner) { st(a.u+row0, x4, ux); st(a.u+a.N+row0, x4, uy); st(a.u+2ull*a.N+row0, x4, uz); }
		}
	} else if (owner && moving_ext && bo != TYPE_E) {
		uint64_t j[Q]; neighbors<Q>(a, x, y, z, j);
		bool moving = false;
		static_for<1, Q, 1>([&](auto I) {
			const uint64_t ji = j[I];
			moving = moving || ((a.flags[ji]&TYPE_BO) == TYPE_S && (a.u[ji] != 0.0f || a.u[a.N+ji] != 0.0f || a.u[2ull*a.N+ji] != 0.0f));
		});
		fl = moving ? (uint8_t)(fl|TYPE_MS) : (uint8_t)(fl&~TYPE_MS);
		st(a.flags+row0, x, fl);
	}
	return true;
}
// One workgroup per lattice row chunk like k_collide (row bases wave-uniform).  store_f with t = 1 (kernel.cpp:1445) puts
// f_eq[i+1] into stream i at the cell itself and f_eq[i] into stream i+1 at the cell's +c_i neighbour, i.e. one column to
// the right or left when c_i has an x component.  A kernel that only stores must not leave memory sectors half written
// (a wave's 64 stores shifted by one element cost 1.7x the time of aligned ones on MI355X), so every lane writes column x
// of every stream: the value for a shifted stream comes from the neighbouring lane through a wave shuffle, and the lanes
// at the two ends of a wave work out the neighbouring cell's equilibrium themselves.  Same bits in the same places.
template <int Q, int ST> __global__ __launch_bounds__(256) void k_initialize(const Args a) {
	using S = Store<ST>; using raw = typename S::raw;
	const uint32_t chunk = blockIdx.x%a.xchunks, y = blockIdx.x/a.xchunks, z = blockIdx.y;
	if ((a.hy && (y == 0u || y >= a.Ny-1u)) || (a.hz && (z == 0u || z >= a.Nz-1u))) return; // kernel.cpp:850-853
	const uint32_t x = chunk*blockDim.x+threadIdx.x, lane = threadIdx.x&63u;
	if (x >= a.Nx) return; // no lane in the row ever takes a value from these
	const uint32_t yq[3] = {y, y+1u == a.Ny ? 0u : y+1u, y == 0u ? a.Ny-1u : y-1u}; // index 0: same, 1: +1, 2: -1
	const uint32_t zq[3] = {z, z+1u == a.Nz ? 0u : z+1u, z == 0u ? a.Nz-1u : z-1u};
	const uint64_t row0 = ((uint64_t)z*a.Ny+y)*a.Nx;
	raw* const fi = static_cast<raw*>(a.fi);
	raw* pa[Q];
	pa[0] = fi+row0;
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		pa[i] = fi+(uint64_t)i*a.S+row0;
		pa[i+1] = fi+(uint64_t)(i+1)*a.S+((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx;
	});
	typedef typename std::conditional<sizeof(raw) == 2, uint32_t, raw>::type lane_t; // what travels through a wave shuffle
	float rho, ux, uy, uz;
	const bool valid = initial_state<Q>(a, x, y, z, row0, true, rho, ux, uy, uz);
	raw e[Q]; // the cell's encoded equilibrium
	{
		float feq[Q];
		f_eq<Q>(valid ? rho : 1.0f, valid ? ux : 0.0f, valid ? uy : 0.0f, valid ? uz : 0.0f, feq);
		static_for<0, Q, 1>([&](auto I) { e[I] = S::encode(feq[I]); });
	}
	// the cell to the left gives the values of the streams with c_x > 0, the cell to the right those with c_x < 0 (periodic)
	const bool seam_lo = lane == 0u, seam_hi = lane == 63u || x+1u == a.Nx;
	bool from_lo = __shfl_up((int)valid, 1) != 0, from_hi = __shfl_down((int)valid, 1) != 0;
	raw shifted[Q];
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0);
		shifted[i+1] = e[i];
		if constexpr (cx > 0) shifted[i+1] = (raw)__shfl_up((lane_t)e[i], 1);
		if constexpr (cx < 0) shifted[i+1] = (raw)__shfl_down((lane_t)e[i], 1);
	});
	auto neighbour_cell = [&](const uint32_t xn, const bool plus, bool& from) { // the lane works out cell xn's equilibrium for the streams it needs
		float r, vx, vy, vz;
		from = initial_state<Q>(a, xn, y, z, row0, false, r, vx, vy, vz);
		if (from) {
			float feq[Q];
			f_eq<Q>(r, vx, vy, vz, feq);
			static_for<1, Q, 2>([&](auto I) {
				constexpr int i = I, cx = dir<Q>(i,0);
				if constexpr (cx != 0) if ((cx > 0) == plus) shifted[i+1] = S::encode(feq[i]);
			});
		}
	};
	if (seam_lo) neighbour_cell(x == 0u ? a.Nx-1u : x-1u, true, from_lo);
	if (seam_hi) neighbour_cell(x+1u == a.Nx ? 0u : x+1u, false, from_hi);
	const uint32_t xb = fresh(x*(uint32_t)sizeof(raw));
	if (valid) {
		st(pa[0], xb, e[0]);
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I;
			st(pa[i], xb, e[i+1]);
			if constexpr (dir<Q>(i,0) == 0) st(pa[i+1], xb, shifted[i+1]);
		});
	}
	if (from_lo) static_for<1, Q, 2>([&](auto I) { constexpr int i = I; if constexpr (dir<Q>(i,0) > 0) st(pa[i+1], xb, shifted[i+1]); });
	if (from_hi) static_for<1, Q, 2>([&](auto I) { constexpr int i = I; if constexpr (dir<Q>(i,0) < 0) st(pa[i+1], xb, shifted[i+1]); });
}

// The same for 16-bit DDFs with two cells per lane (rows of an even number of cells): every stream leaves as aligned dwords,
// 256 bytes per wave instruction instead of 128.  Column x0 of a stream with c_x > 0 takes the value of the previous lane's
// second cell, column x0+1 that of the lane's first cell; with c_x < 0 it is the lane's second cell and the next lane's first.
template <int Q, int ST> __global__ __launch_bounds__(256) void k_initialize_pairs(const Args a) {
	using S = Store<ST>; using raw = typename S::raw;
	static_assert(sizeof(raw) == 2, "16-bit DDF storage");
	const uint32_t chunk = blockIdx.x%a.xchunks, y = blockIdx.x/a.xchunks, z = blockIdx.y;
	if ((a.hy && (y == 0u || y >= a.Ny-1u)) || (a.hz && (z == 0u || z >= a.Nz-1u))) return; // kernel.cpp:850-853
	const uint32_t x0 = (chunk*blockDim.x+threadIdx.x)*2u, lane = threadIdx.x&63u;
	if (x0 >= a.Nx) return;
	const uint32_t yq[3] = {y, y+1u == a.Ny ? 0u : y+1u, y == 0u ? a.Ny-1u : y-1u}; // index 0: same, 1: +1, 2: -1
	const uint32_t zq[3] = {z, z+1u == a.Nz ? 0u : z+1u, z == 0u ? a.Nz-1u : z-1u};
	const uint64_t row0 = ((uint64_t)z*a.Ny+y)*a.Nx;
	raw* const fi = static_cast<raw*>(a.fi);
	raw* pa[Q];
	pa[0] = fi+row0;
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		pa[i] = fi+(uint64_t)i*a.S+row0;
		pa[i+1] = fi+(uint64_t)(i+1)*a.S+((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx;
	});
	uint32_t e[Q]; // the two cells' encoded equilibria, the first cell in the low half
	bool ok[2];
#pragma unroll
	for (int k = 0; k < 2; k++) {
		float rho, ux, uy, uz;
		ok[k] = initial_state<Q>(a, x0+(uint32_t)k, y, z, row0, true, rho, ux, uy, uz);
		float feq[Q];
		f_eq<Q>(ok[k] ? rho : 1.0f, ok[k] ? ux : 0.0f, ok[k] ? uy : 0.0f, ok[k] ? uz : 0.0f, feq);
		static_for<0, Q, 1>([&](auto I) { const uint32_t c = S::encode(feq[I]); e[I] = k == 0 ? c : (e[I]|(c<<16)); });
	}
	const bool seam_lo = lane == 0u, seam_hi = lane == 63u || x0+2u >= a.Nx;
	bool from_lo = __shfl_up((int)ok[1], 1) != 0, from_hi = __shfl_down((int)ok[0], 1) != 0;
	uint32_t other[Q]; // for the x-shifted streams: the neighbouring lane's cell that lands in this lane's columns
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0);
		other[i+1] = 0u;
		if constexpr (cx > 0) other[i+1] = __shfl_up(e[i], 1)>>16;        // previous lane's second cell
		if constexpr (cx < 0) other[i+1] = __shfl_down(e[i], 1)&0xFFFFu; // next lane's first cell
	});
	auto neighbour_cell = [&](const uint32_t xn, const bool plus, bool& from) { // the lane works out cell xn's equilibrium for the streams it needs
		float r, vx, vy, vz;
		from = initial_state<Q>(a, xn, y, z, row0, false, r, vx, vy, vz);
		if (from) {
			float feq[Q];
			f_eq<Q>(r, vx, vy, vz, feq);
			static_for<1, Q, 2>([&](auto I) {
				constexpr int i = I, cx = dir<Q>(i,0);
				if constexpr (cx != 0) if ((cx > 0) == plus) other[i+1] = S::encode(feq[i]);
			});
		}
	};
	if (seam_lo) neighbour_cell(x0 == 0u ? a.Nx-1u : x0-1u, true, from_lo);
	if (seam_hi) neighbour_cell(x0+2u >= a.Nx ? 0u : x0+2u, false, from_hi);
	const uint32_t xb = fresh(x0*2u);
	auto put = [&](raw* base, const uint32_t lo, const bool lo_ok, const uint32_t hi, const bool hi_ok) { // columns x0 and x0+1 of one stream
		if (lo_ok && hi_ok) st(base, xb, (lo&0xFFFFu)|(hi<<16));
		else if (lo_ok) st(base, xb, (raw)lo);
		else if (hi_ok) st(base, xb+2u, (raw)hi);
	};
	put(pa[0], e[0], ok[0], e[0]>>16, ok[1]);
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0);
		put(pa[i], e[i+1], ok[0], e[i+1]>>16, ok[1]);
		if constexpr (cx == 0) put(pa[i+1], e[i], ok[0], e[i]>>16, ok[1]);
		if constexpr (cx > 0) put(pa[i+1], other[i+1], from_lo, e[i], ok[0]);       // cells x0-1 and x0
		if constexpr (cx < 0) put(pa[i+1], e[i]>>16, ok[1], other[i+1], from_hi);   // cells x0+1 and x0+2
	});
}

// FluidX3D-master/src/kernel.cpp:1897-1910
template <int Q, int ST> __global__ __launch_bounds__(256) void k_update_force_field(const Args a) {
	const Cell c = cell_of_thread(a);
	if (!c.valid || is_halo(a, c)) return;
	const uint64_t n = c.n;
	if ((a.flags[n]&TYPE_BO) != TYPE_S) return;
	uint64_t j[Q]; neighbors<Q>(a, c.x, c.y, c.z, j);
	float f[Q]; load_f<Q, ST>(a, n, j, a.odd, f);
	float r, ax, ay, az; rho_u<Q>(f, r, ax, ay, az);
	a.F[n] = 2.0f*ax*r; a.F[a.N+n] = 2.0f*ay*r; a.F[2ull*a.N+n] = 2.0f*az*r;
}
// FluidX3D-master/src/kernel.cpp:1911-1917 (runs on the halo too)
__global__ __launch_bounds__(256) void k_reset_force_field(const Args a) {
	const Cell c = cell_of_thread(a);
	if (!c.valid) return;
	const uint64_t n = c.n;
	a.F[n] = 0.0f; a.F[a.N+n] = 0.0f; a.F[2ull*a.N+n] = 0.0f;
}
// FluidX3D-master/src/kernel.cpp:1449-1465
template <int Q> __global__ __launch_bounds__(256) void k_update_moving_boundaries(const Args a) {
	const Cell c = cell_of_thread(a);
	if (!c.valid || is_halo(a, c)) return;
	const uint64_t n = c.n;
	const uint8_t fl = a.flags[n], bo = fl&TYPE_BO;
	if (bo == TYPE_S || bo == TYPE_E || (fl&TYPE_T)) return;
	uint64_t j[Q]; neighbors<Q>(a, c.x, c.y, c.z, j);
	bool moving = false;
	static_for<1, Q, 1>([&](auto I) {
		const uint64_t ji = j[I];
		moving = moving || ((a.u[ji] != 0.0f || a.u[a.N+ji] != 0.0f || a.u[2ull*a.N+ji] != 0.0f) && (a.flags[ji]&TYPE_BO) == TYPE_S);
	});
	a.flags[n] = moving ? (uint8_t)(fl|TYPE_MS) : (uint8_t)(fl&~TYPE_MS);
}

// ---------------------------------------------------------------------------------------------- halo pack / unpack
// FluidX3D-master/src/kernel.cpp:2077-2223.  One thread per face cell a and side; buffer layout buf[b*A+a].
struct HaloArgs {
	Args lat;
	void* buf[2];   // [0]: + side, [1]: - side
	uint32_t axis;  // 0=x 1=y 2=z
	uint32_t A;     // face area
	uint32_t insert;
};
__device__ __forceinline__ void face_cell(const Args& a, uint32_t axis, uint32_t fa, uint32_t layer, uint32_t& x, uint32_t& y, uint32_t& z) { // kernel.cpp:2081-2096
	if (axis == 0u) { x = layer; y = fa%a.Ny; z = fa/a.Ny; }
	else if (axis == 1u) { x = fa/a.Nz; y = layer; z = fa%a.Nz; }
	else { x = fa%a.Nx; y = fa/a.Nx; z = layer; }
}
template <int Q, int ST> __global__ __launch_bounds__(256) void k_halo_fi(const HaloArgs h) {
	using raw = typename Store<ST>::raw;
	const Args& a = h.lat;
	const uint32_t fa = blockIdx.x*blockDim.x+threadIdx.x, side = blockIdx.y;
	if (fa >= h.A) return;
	const uint32_t Nax = h.axis == 0u ? a.Nx : h.axis == 1u ? a.Ny : a.Nz;
	const uint32_t layer = h.insert ? (side == 0u ? Nax-1u : 0u) : (side == 0u ? Nax-2u : 1u);
	uint32_t x, y, z; face_cell(a, h.axis, fa, layer, x, y, z);
	const uint64_t n = (uint64_t)x+((uint64_t)y+(uint64_t)z*a.Ny)*a.Nx;
	uint64_t j[Q]; neighbors<Q>(a, x, y, z, j);
	raw* fi = static_cast<raw*>(a.fi);
	raw* buf = static_cast<raw*>(h.buf[side]);
	constexpr int T = transfers<Q>();
	static_for<0, 3, 1>([&](auto AX) {
		constexpr int ax = AX;
		if constexpr (ax < (Q == 9 ? 2 : 3)) {
			if (h.axis == (uint32_t)ax) {
				static_for<0, T, 1>([&](auto B) {
					constexpr int b = B, P = transfer_dir<Q>(ax, b), M = (P%2) ? P+1 : P-1; // + side sends P, - side its opposite M
					const uint64_t k = (uint64_t)b*h.A+fa;
					auto move = [&](auto II) {
						constexpr int i = II, opp = (i%2) ? i+1 : i-1;
						if (!h.insert) buf[k] = fi[(uint64_t)(a.odd ? opp : i)*a.S+((i%2) ? j[i] : n)];                 // kernel.cpp:2135
						else fi[(uint64_t)(a.odd ? i : opp)*a.S+((i%2) ? n : j[(i%2) ? i : i-1])] = buf[k];            // kernel.cpp:2144
					};
					if (side == 0u) move(std::integral_constant<int, P>{}); else move(std::integral_constant<int, M>{});
				});
			}
		}
	});
}
// what: 1 = rho,u,flags (17 B per face cell, kernel.cpp:2161-2186); 2 = flags (kernel.cpp:2188-2199); 3 = F (kernel.cpp:2202-2223)
__global__ __launch_bounds__(256) void k_halo_fields(const HaloArgs h, const uint32_t what) {
	const Args& a = h.lat;
	const uint32_t fa = blockIdx.x*blockDim.x+threadIdx.x, side = blockIdx.y;
	if (fa >= h.A) return;
	const uint32_t Nax = h.axis == 0u ? a.Nx : h.axis == 1u ? a.Ny : a.Nz;
	const uint32_t layer = h.insert ? (side == 0u ? Nax-1u : 0u) : (side == 0u ? Nax-2u : 1u);
	uint32_t x, y, z; face_cell(a, h.axis, fa, layer, x, y, z);
	const uint64_t n = (uint64_t)x+((uint64_t)y+(uint64_t)z*a.Ny)*a.Nx, A = h.A;
	float* bf = static_cast<float*>(h.buf[side]);
	uint8_t* bb = static_cast<uint8_t*>(h.buf[side]);
	if (what == 1u) {
		if (!h.insert) { bf[fa] = a.rho[n]; bf[A+fa] = a.u[n]; bf[2u*A+fa] = a.u[a.N+n]; bf[3u*A+fa] = a.u[2ull*a.N+n]; bb[16u*A+fa] = a.flags[n]; }
		else { a.rho[n] = bf[fa]; a.u[n] = bf[A+fa]; a.u[a.N+n] = bf[2u*A+fa]; a.u[2ull*a.N+n] = bf[3u*A+fa]; a.flags[n] = bb[16u*A+fa]; }
	} else if (what == 2u) {
		if (!h.insert) bb[fa] = a.flags[n]; else a.flags[n] = bb[fa];
	} else {
		if (!h.insert) { bf[fa] = a.F[n]; bf[A+fa] = a.F[a.N+n]; bf[2u*A+fa] = a.F[2ull*a.N+n]; }
		else { a.F[n] = bf[fa]; a.F[a.N+n] = bf[A+fa]; a.F[2ull*a.N+n] = bf[2u*A+fa]; }
	}
}

// ---------------------------------------------------------------------------------------------- mesh voxelizer
// FluidX3D-master/src/kernel.cpp:2295-2390.  One lane per cell of the face perpendicular to `direction`: a ray along
// the axis, bidirectional Moeller-Trumbore against every triangle, the hits ahead kept as truncated 16-bit distances
// and sorted, then the column is walked and the cells between an odd and the next even hit are flagged.  Plain float
// arithmetic in a fixed order (dot = (ax*bx+ay*by)+az*bz) so that the flag lattice is reproducible bit for bit.
struct VoxelArgs {
	Args lat;
	const float* p0; const float* p1; const float* p2; // 3 floats per triangle each
	const float4* boxes; // per triangle: its extent in the two coordinates across the rays (see k_triangle_boxes), or nullptr: every ray tests every triangle
	uint32_t triangles, direction;
	int32_t O[3];
	float lo[3], hi[3];  // bounding box of the mesh, widened by 2 cells (lbm.cpp:352-358)
	float rc[3], lin[3], rot[3];
	uint8_t flag;
};
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3sub(V3 a, V3 b) { return {a.x-b.x, a.y-b.y, a.z-b.z}; }
__device__ __forceinline__ V3 v3add(V3 a, V3 b) { return {a.x+b.x, a.y+b.y, a.z+b.z}; }
__device__ __forceinline__ float v3dot(V3 a, V3 b) { return a.x*b.x+a.y*b.y+a.z*b.z; }
__device__ __forceinline__ V3 v3cross(V3 a, V3 b) { return {a.y*b.z-a.z*b.y, a.z*b.x-a.x*b.z, a.x*b.y-a.y*b.x}; }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : x > hi ? hi : x; }
__device__ __forceinline__ V3 cell_position(const VoxelArgs& v, uint32_t x, uint32_t y, uint32_t z) { // kernel.cpp:834-836, 2306
	const Args& a = v.lat;
	const V3 p = {(float)x+0.5f-0.5f*(float)a.Nx, (float)y+0.5f-0.5f*(float)a.Ny, (float)z+0.5f-0.5f*(float)a.Nz};
	const V3 off = {0.5f*(float)((int)a.Nx+2*v.O[0])-0.5f, 0.5f*(float)((int)a.Ny+2*v.O[1])-0.5f, 0.5f*(float)((int)a.Nz+2*v.O[2])-0.5f};
	return v3add(p, off);
}
// The reference lets every ray test every triangle (kernel.cpp:2313-2328).  The test can only succeed for a ray that passes
// through the triangle's projection, so a wave may skip a triangle whose projection lies clear of all its 64 rays - the
// triangles that remain are visited in the same order with the same arithmetic, hence the same flags.  k_triangle_boxes
// writes, per triangle, the extent of its projection in the two coordinates across the rays, widened by half a cell (the
// rays are one cell apart; the rounding error of the intersection test is orders of magnitude smaller for the triangles
// that get a box).  A triangle seen at less than ~3 degrees by the rays, or a sliver, gets an unbounded box and is never
// skipped: there the computed barycentric coordinates are dominated by rounding and may accept rays far from the projection,
// which the result must reproduce.
__global__ __launch_bounds__(256) void k_triangle_boxes(const VoxelArgs v, float4* boxes) {
	const uint32_t i = blockIdx.x*blockDim.x+threadIdx.x;
	if (i >= v.triangles) return;
	const V3 a0 = {v.p0[3u*i], v.p0[3u*i+1u], v.p0[3u*i+2u]}, a1 = {v.p1[3u*i], v.p1[3u*i+1u], v.p1[3u*i+2u]}, a2 = {v.p2[3u*i], v.p2[3u*i+1u], v.p2[3u*i+2u]};
	const uint32_t dirn = v.direction;
	const V3 dirv = {(float)(dirn == 0u), (float)(dirn == 1u), (float)(dirn == 2u)};
	const V3 e1 = v3sub(a1, a0), e2 = v3sub(a2, a0);
	const float g = v3dot(e1, v3cross(dirv, e2));
	const float scale = sqrtf(v3dot(e1, e1))*sqrtf(v3dot(e2, e2));
	const float inf = __builtin_huge_valf();
	float4 box = {-inf, inf, -inf, inf};
	if (fabsf(g) >= 0.05f*scale && scale < inf) { // well-conditioned; anything else (NaN included) keeps the unbounded box
		const float pa[3] = {dirn == 0u ? a0.y : a0.x, dirn == 0u ? a1.y : a1.x, dirn == 0u ? a2.y : a2.x}; // first coordinate across the rays
		const float pb[3] = {dirn == 2u ? a0.y : a0.z, dirn == 2u ? a1.y : a1.z, dirn == 2u ? a2.y : a2.z}; // second
		box.x = fminf(fminf(pa[0], pa[1]), pa[2])-0.5f; box.y = fmaxf(fmaxf(pa[0], pa[1]), pa[2])+0.5f;
		box.z = fminf(fminf(pb[0], pb[1]), pb[2])-0.5f; box.w = fmaxf(fmaxf(pb[0], pb[1]), pb[2])+0.5f;
	}
	boxes[i] = box;
}
__device__ __forceinline__ float wave_min(float x) { for (int m = 32; m > 0; m >>= 1) x = fminf(x, __shfl_xor(x, m)); return x; }
__device__ __forceinline__ float wave_max(float x) { for (int m = 32; m > 0; m >>= 1) x = fmaxf(x, __shfl_xor(x, m)); return x; }
template <int Q, int ST> __global__ __launch_bounds__(64) void k_voxelize_mesh(const VoxelArgs v) {
	using S = Store<ST>; using raw = typename S::raw;
	const Args& a = v.lat;
	const uint32_t dirn = v.direction;
	const uint32_t A = dirn == 0u ? a.Ny*a.Nz : dirn == 1u ? a.Nz*a.Nx : a.Nx*a.Ny;
	const uint32_t fa_raw = blockIdx.x*blockDim.x+threadIdx.x;
	const bool in_face = fa_raw < A;
	const uint32_t fa = in_face ? fa_raw : A-1u; // lanes past the face stay for the wave reductions below and leave after them
	const int Nd[3] = {(int)a.Nx, (int)a.Ny, (int)a.Nz};
	uint32_t c[3];
	const uint32_t start = (uint32_t)clampi((int)v.lo[dirn]-v.O[dirn], 0, Nd[dirn]-1);
	if (dirn == 0u) { c[0] = start; c[1] = fa%a.Ny; c[2] = fa/a.Ny; }
	else if (dirn == 1u) { c[0] = fa/a.Nz; c[1] = start; c[2] = fa%a.Nz; }
	else { c[0] = fa%a.Nx; c[1] = fa/a.Nx; c[2] = start; }
	const V3 origin = cell_position(v, c[0], c[1], c[2]);
	const V3 dirv = {(float)(dirn == 0u), (float)(dirn == 1u), (float)(dirn == 2u)};
	const bool outside = dirn == 0u ? (origin.y < v.lo[1] || origin.z < v.lo[2] || origin.y >= v.hi[1] || origin.z >= v.hi[2])
	                   : dirn == 1u ? (origin.x < v.lo[0] || origin.z < v.lo[2] || origin.x >= v.hi[0] || origin.z >= v.hi[2])
	                   : (origin.x < v.lo[0] || origin.y < v.lo[1] || origin.x >= v.hi[0] || origin.y >= v.hi[1]);
	// the extent of this wave's rays in the two coordinates across them (only rays that take part count)
	const bool active = in_face && !outside;
	const float ra = dirn == 0u ? origin.y : origin.x, rb = dirn == 2u ? origin.y : origin.z;
	const float big = __builtin_huge_valf();
	const float wa0 = wave_min(active ? ra : big), wa1 = wave_max(active ? ra : -big), wb0 = wave_min(active ? rb : big), wb1 = wave_max(active ? rb : -big);
	if (__ballot(active) == 0ull) return;
	uint32_t ahead = 0u, behind = 0u;
	uint16_t dist[64];
	for (uint32_t i = 0u; i < 64u; i++) dist[i] = 0u;
	// Triangles in batches of 64: every lane (rays that do not take part included) fetches one box and compares it with the
	// wave's extent, the votes form the list of triangles this wave has to test, and those are visited in ascending order.
	const float4 nowhere = {big, -big, big, -big}; // the box of a triangle that is not there
	float4 ahead_box = (v.boxes && threadIdx.x < v.triangles) ? v.boxes[threadIdx.x] : nowhere; // fetched one batch ahead of its use
	for (uint32_t i0 = 0u; i0 < v.triangles; i0 += 64u) {
		const uint32_t left = v.triangles-i0;
		uint64_t todo = left >= 64u ? ~0ull : (1ull<<left)-1ull;
		if (v.boxes) {
			const float4 box = ahead_box;
			ahead_box = (i0+64u+threadIdx.x < v.triangles) ? v.boxes[i0+64u+threadIdx.x] : nowhere;
			todo = __ballot(!(box.y < wa0 || box.x > wa1 || box.w < wb0 || box.z > wb1));
		}
		while (todo != 0ull) {
			const uint32_t i = i0+(uint32_t)__builtin_ctzll(todo);
			todo &= todo-1ull;
			if (!active) continue;
			const V3 a0 = {v.p0[3u*i], v.p0[3u*i+1u], v.p0[3u*i+2u]}, a1 = {v.p1[3u*i], v.p1[3u*i+1u], v.p1[3u*i+2u]}, a2 = {v.p2[3u*i], v.p2[3u*i+1u], v.p2[3u*i+2u]};
			const V3 e1 = v3sub(a1, a0), e2 = v3sub(a2, a0), w = v3sub(origin, a0), h = v3cross(dirv, e2), q = v3cross(w, e1);
			const float g = v3dot(e1, h), f = 1.0f/g, s = f*v3dot(w, h), tt = f*v3dot(dirv, q), d = f*v3dot(e2, q);
			if (g != 0.0f && s >= 0.0f && s < 1.0f && tt >= 0.0f && s+tt < 1.0f) {
				if (d > 0.0f) { if (ahead < 64u && d < 65536.0f) dist[ahead] = (uint16_t)d; ahead++; }
				else behind++;
			}
		}
	}
	if (!active) return;
	const uint32_t kept = ahead < 64u ? ahead : 64u;
	for (uint32_t i = 1u; i < kept; i++) { // insertion sort, ascending
		const uint16_t val = dist[i]; uint32_t j = i;
		while (j > 0u && dist[j-1u] > val) { dist[j] = dist[j-1u]; j--; }
		dist[j] = val;
	}
	bool inside = (ahead%2u) && (behind%2u);
	const V3 lin = {v.lin[0], v.lin[1], v.lin[2]}, rot = {v.rot[0], v.rot[1], v.rot[2]}, rc = {v.rc[0], v.rc[1], v.rc[2]};
	const bool set_u = lin.x*lin.x+lin.y*lin.y+lin.z*lin.z+rot.x*rot.x+rot.y*rot.y+rot.z*rot.z > 0.0f;
	uint32_t next = (ahead%2u) != (behind%2u);
	const uint32_t h0 = c[dirn];
	const uint32_t hmax = (uint32_t)clampi((int)v.hi[dirn]-v.O[dirn], 0, Nd[dirn]);
	const uint32_t last = ahead-1u < 63u ? ahead-1u : 63u;
	const uint32_t hmesh = h0+(uint32_t)dist[last];
	for (uint32_t hc = h0; hc < hmax; hc++) {
		while (next < ahead && hc > h0+(uint32_t)dist[next < 63u ? next : 63u]) { inside = !inside; next++; }
		inside = inside && (next < ahead && hc < hmesh);
		uint32_t cc[3] = {c[0], c[1], c[2]}; cc[dirn] = hc;
		const uint64_t n = (uint64_t)cc[0]+((uint64_t)cc[1]+(uint64_t)cc[2]*a.Ny)*a.Nx;
		uint8_t fl = a.flags[n];
		const V3 p = cell_position(v, cc[0], cc[1], cc[2]);
		const V3 u_set = v3add(lin, v3cross(v3sub(rc, p), rot));
		if (inside) {
			fl = (uint8_t)((fl&~TYPE_BO)|v.flag);
			if (set_u) { a.u[n] = u_set.x; a.u[a.N+n] = u_set.y; a.u[2ull*a.N+n] = u_set.z; }
		} else if ((fl&TYPE_BO) == TYPE_S) {
			const float ux = a.u[n], uy = a.u[a.N+n], uz = a.u[2ull*a.N+n];
			if (ux == u_set.x && uy == u_set.y && uz == u_set.z) { // the cell belonged to this mesh: release it
				if (set_u) { // give the freed cell equilibrium DDFs at rho=1 (kernel.cpp:2367-2373)
					uint64_t j[Q]; neighbors<Q>(a, cc[0], cc[1], cc[2], j);
					float feq[Q]; f_eq<Q>(1.0f, ux, uy, uz, feq);
					raw* fi = static_cast<raw*>(a.fi);
					fi[n] = S::encode(feq[0]);
					static_for<1, Q, 2>([&](auto I) {
						constexpr int i = I;
						fi[(uint64_t)(a.odd ? i+1 : i)*a.S+j[i]] = S::encode(feq[i]);
						fi[(uint64_t)(a.odd ? i : i+1)*a.S+n] = S::encode(feq[i+1]);
					});
				}
				fl = (fl&TYPE_BO) == TYPE_MS ? (uint8_t)(fl&~TYPE_MS) : (uint8_t)(fl&~v.flag);
			}
		}
		a.flags[n] = fl;
	}
}
// FluidX3D-master/src/kernel.cpp:2386-2390
__global__ __launch_bounds__(256) void k_unvoxelize_mesh(const VoxelArgs v) {
	const Cell c = cell_of_thread(v.lat);
	if (!c.valid) return;
	const V3 p = cell_position(v, c.x, c.y, c.z);
	if (p.x >= v.lo[0]-1.0f && p.y >= v.lo[1]-1.0f && p.z >= v.lo[2]-1.0f && p.x <= v.hi[0]+1.0f && p.y <= v.hi[1]+1.0f && p.z <= v.hi[2]+1.0f) v.lat.flags[c.n] &= (uint8_t)~v.flag;
}

// ---------------------------------------------------------------------------------------------- object sums
// Sums over all cells whose flags equal `marker` (FluidX3D-master/src/kernel.cpp:1929-1987): what = 0 positions and cell
// count (centre of mass), 1 boundary force F, 2 torque (r - centre) x F.  The reference reduces 64 cells in LDS and adds
// the group sums with float atomics, so its result depends on the order the atomics land; here every workgroup writes
// one partial sum (wave shuffles, then LDS across the 4 waves) and the host adds the partials in double: reproducible.
// skip_halo = 0 is the reference kernel's own definition for one domain (every cell of the array, positions from the
// domain's own centre); skip_halo = 1 leaves halo layers to their owners and measures from the centre of the global box.
struct SumArgs { Args lat; float4* partial; uint32_t what; uint8_t marker; uint8_t skip_halo; float hx, hy, hz, cx, cy, cz; int32_t O[3]; };
__global__ __launch_bounds__(256) void k_object_sum(const SumArgs s) {
	const Args& a = s.lat;
	float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
	// a workgroup takes whole lattice rows (row = y+z*Ny, wave-uniform), its lanes the cells of the row: no per-cell division
	const uint32_t rows = a.Ny*a.Nz;
	for (uint32_t row = blockIdx.x; row < rows; row += gridDim.x) {
		const uint32_t z = row/a.Ny, y = row-z*a.Ny;
		if (s.skip_halo && ((a.hy && (y == 0u || y == a.Ny-1u)) || (a.hz && (z == 0u || z == a.Nz-1u)))) continue;
		const uint64_t row0 = (uint64_t)row*a.Nx;
		// position() of kernel.cpp:834-836: coordinate + 0.5 - 0.5*size, then minus the rotation centre (kernel.cpp:1975)
		const float py = (float)((int)y+s.O[1])+0.5f-s.hy-s.cy, pz = (float)((int)z+s.O[2])+0.5f-s.hz-s.cz;
		for (uint32_t x = threadIdx.x; x < a.Nx; x += blockDim.x) {
			if ((s.skip_halo && a.hx && (x == 0u || x == a.Nx-1u)) || ld<uint8_t>(a.flags+row0, x) != s.marker) continue;
			const float px = (float)((int)x+s.O[0])+0.5f-s.hx-s.cx;
			if (s.what == 0u) { acc.x += px; acc.y += py; acc.z += pz; acc.w += 1.0f; }
			else {
				const float fx = ld<float>(a.F+row0, 4u*x), fy = ld<float>(a.F+a.N+row0, 4u*x), fz = ld<float>(a.F+2ull*a.N+row0, 4u*x);
				if (s.what == 1u) { acc.x += fx; acc.y += fy; acc.z += fz; }
				else { acc.x += py*fz-pz*fy; acc.y += pz*fx-px*fz; acc.z += px*fy-py*fx; }
				acc.w += 1.0f;
			}
		}
	}
	for (int off = 32; off > 0; off >>= 1) { acc.x += __shfl_down(acc.x, off); acc.y += __shfl_down(acc.y, off); acc.z += __shfl_down(acc.z, off); acc.w += __shfl_down(acc.w, off); }
	__shared__ float4 wave[4];
	if ((threadIdx.x&63u) == 0u) wave[threadIdx.x>>6] = acc;
	__syncthreads();
	if (threadIdx.x == 0u) {
		float4 t = wave[0];
		for (int w = 1; w < 4; w++) { t.x += wave[w].x; t.y += wave[w].y; t.z += wave[w].z; t.w += wave[w].w; }
		s.partial[blockIdx.x] = t;
	}
}

#endif // LBM_SMALL_KERNELS

} // namespace lbm
