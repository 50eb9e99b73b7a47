// lbm_kernels.hpp - the lattice kernels (templates) and their launchers.
//
// Thread mapping of the fused stream+collide kernel: one workgroup works on one lattice row (fixed y,z),
// or on an x-chunk of it; y, z and every neighbour-row base are therefore wave-uniform (SGPRs) and only
// the x offset is per lane.  Each lane handles V consecutive cells so that the x-aligned DDF slots move as
// 4*V (FP32) or 2*V (FP16) byte accesses.  Every DDF slot is read and written by exactly one cell per
// step (esoteric-pull, FluidX3D-master/src/kernel.cpp:1332-1345), so there is no inter-thread hazard and
// the traffic is the algorithmic 2*Q*sizeof(fpxx) bytes per cell.
#pragma once
#include "lbm_device.hpp"

namespace lbm {

template <class T, int V> struct VecOf { typedef T type __attribute__((ext_vector_type(V))); };
template <class T> struct VecOf<T, 1> { typedef T type; };

// V consecutive raw DDFs starting at p[x]; ALIGNED says x*sizeof(raw) is a multiple of the vector size
template <class raw, int V, bool ALIGNED> __device__ __forceinline__ void load_raw(const raw* __restrict__ p, uint32_t x, raw* out) {
	if constexpr (V == 1) out[0] = p[x];
	else if constexpr (ALIGNED) {
		typedef typename VecOf<raw, V>::type vec;
		const vec t = *reinterpret_cast<const vec*>(p+x);
		for (int v = 0; v < V; v++) out[v] = t[v];
	} else {
		typedef typename VecOf<raw, V>::type vec;
		vec t; __builtin_memcpy(&t, p+x, sizeof(vec));
		for (int v = 0; v < V; v++) out[v] = t[v];
	}
}
template <class raw, int V, bool ALIGNED> __device__ __forceinline__ void store_raw(raw* __restrict__ p, uint32_t x, const raw* in) {
	if constexpr (V == 1) p[x] = in[0];
	else {
		typedef typename VecOf<raw, V>::type vec;
		vec t;
		for (int v = 0; v < V; v++) t[v] = in[v];
		if constexpr (ALIGNED) *reinterpret_cast<vec*>(p+x) = t; else __builtin_memcpy(p+x, &t, sizeof(vec));
	}
}

// ---------------------------------------------------------------------------------------------- stream_collide / update_fields
// FluidX3D-master/src/kernel.cpp:1470-1654 (FIELDS_ONLY=false) and 1812-1894 (FIELDS_ONLY=true).
// FEAT >= 0: features fixed at compile time; FEAT < 0: taken from Args::features.
template <int Q, int ST, int COLL, int FEAT, int V, bool FIELDS_ONLY>
__global__ __launch_bounds__(256) void k_collide(const Args a) {
	using S = Store<ST>;
	using raw = typename S::raw;
	// grid = (xchunks*Ny, planes): a launch may not exceed 2^32 work-items per grid dimension, and 1600^3 lattices do
	const uint32_t chunk = blockIdx.x%a.xchunks, y = blockIdx.x/a.xchunks, z = a.z0+blockIdx.y;
	if (a.hy && (y == 0u || y == a.Ny-1u)) return;
	const uint32_t x = (chunk*blockDim.x+threadIdx.x)*V;
	if (x >= a.Nx) return;

	// wave-uniform row bases of the 3x3 neighbourhood in (y,z), periodic on the local lattice (kernel.cpp:914-925)
	const uint32_t yq[3] = {y, y+1u == a.Ny ? 0u : y+1u, y == 0u ? a.Ny-1u : y-1u}; // index 0: same, 1: +1, 2: -1
	const uint32_t zq[3] = {z, z+1u == a.Nz ? 0u : z+1u, z == 0u ? a.Nz-1u : z-1u};
	const uint64_t row0 = ((uint64_t)z*a.Ny+y)*a.Nx;
	raw* const fi = static_cast<raw*>(a.fi);

	// which of the V cells take part (solid cells and halo cells are skipped, kernel.cpp:1482-1485)
	uint8_t flagsn[V];
	bool active[V];
	bool any = false;
	if constexpr (V == 1) flagsn[0] = a.flags[row0+x];
	else load_raw<uint8_t, V, true>(a.flags+row0, x, flagsn);
#pragma unroll
	for (int v = 0; v < V; v++) {
		const uint32_t xv = x+(uint32_t)v;
		active[v] = (flagsn[v]&TYPE_BO) != TYPE_S && !(a.hx && (xv == 0u || xv == a.Nx-1u));
		any = any || active[v];
	}
	if (!any) return;

	auto feat = [&]() { if constexpr (FEAT >= 0) return StaticFeat<(uint32_t)(FEAT >= 0 ? FEAT : 0)>{}; else return DynamicFeat{a.features}; }();

	// ---- streaming part 2: gather the DDFs of the V cells (kernel.cpp:1332-1338)
	float f[V][Q];
	{
		raw t[V];
		load_raw<raw, V, true>(fi+row0, x, t);
#pragma unroll
		for (int v = 0; v < V; v++) f[v][0] = S::decode(t[v]);
	}
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0), cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		const uint32_t sA = a.odd ? i : i+1, sB = a.odd ? i+1 : i;
		raw t[V];
		load_raw<raw, V, true>(fi+(uint64_t)sA*a.N+row0, x, t);
#pragma unroll
		for (int v = 0; v < V; v++) f[v][i] = S::decode(t[v]);
		const raw* pn = fi+(uint64_t)sB*a.N+((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx;
		if constexpr (cx == 0) load_raw<raw, V, true>(pn, x, t);
		else if constexpr (cx > 0) {
			if (x+(uint32_t)V < a.Nx) load_raw<raw, V, false>(pn, x+1u, t);
			else {
#pragma unroll
				for (int v = 0; v < V; v++) { const uint32_t xn = x+(uint32_t)v+1u; t[v] = pn[xn == a.Nx ? 0u : xn]; }
			}
		} else {
			if (x > 0u) load_raw<raw, V, false>(pn, x-1u, t);
			else {
#pragma unroll
				for (int v = 0; v < V; v++) { const uint32_t xv = x+(uint32_t)v; t[v] = pn[xv == 0u ? a.Nx-1u : xv-1u]; }
			}
		}
#pragma unroll
		for (int v = 0; v < V; v++) f[v][i+1] = S::decode(t[v]);
	});

	// ---- per cell: boundaries, moments, collision
	float rhon[V], uxn[V], uyn[V], uzn[V];
	bool wf[V];
#pragma unroll
	for (int v = 0; v < V; v++) {
		const uint32_t xv = x+(uint32_t)v;
		const uint64_t n = row0+xv;
		const uint8_t bo = flagsn[v]&TYPE_BO;
		wf[v] = false;
		if (!active[v]) continue;
		if (feat(F_MOVING_BOUNDARIES) && bo == TYPE_MS) { // kernel.cpp:1111-1118
			const uint32_t xq[3] = {xv, xv+1u == a.Nx ? 0u : xv+1u, xv == 0u ? a.Nx-1u : xv-1u};
			static_for<1, Q, 2>([&](auto I) {
				constexpr int i = I;
				constexpr float w6 = -6.0f*weight<Q>(i);
				auto nb = [&](auto J) -> uint64_t {
					constexpr int jx = dir<Q>(J,0), jy = dir<Q>(J,1), jz = dir<Q>(J,2);
					return ((uint64_t)zq[jz == 0 ? 0 : jz > 0 ? 1 : 2]*a.Ny+yq[jy == 0 ? 0 : jy > 0 ? 1 : 2])*a.Nx+xq[jx == 0 ? 0 : jx > 0 ? 1 : 2];
				};
				uint64_t ji = nb(std::integral_constant<int, i+1>{});
				if ((a.flags[ji]&TYPE_BO) == TYPE_S) f[v][i] = fmaf(w6, cdot<Q, i+1>(a.u[ji], a.u[a.N+ji], a.u[2ull*a.N+ji]), f[v][i]);
				ji = nb(std::integral_constant<int, i>{});
				if ((a.flags[ji]&TYPE_BO) == TYPE_S) f[v][i+1] = fmaf(w6, cdot<Q, i>(a.u[ji], a.u[a.N+ji], a.u[2ull*a.N+ji]), f[v][i+1]);
			});
		}
		collide<Q, COLL, FIELDS_ONLY>(f[v], feat, bo, a, n, rhon[v], uxn[v], uyn[v], uzn[v], wf[v]);
	}

	// ---- rho/u output (kernel.cpp:1581-1591, 1881-1893)
	if (FIELDS_ONLY || feat(F_UPDATE_FIELDS)) {
#pragma unroll
		for (int v = 0; v < V; v++) if (wf[v]) {
			const uint64_t n = row0+x+(uint32_t)v;
			a.rho[n] = rhon[v]; a.u[n] = uxn[v]; a.u[a.N+n] = uyn[v]; a.u[2ull*a.N+n] = uzn[v];
		}
	}
	if constexpr (FIELDS_ONLY) return;

	// ---- streaming part 1: scatter to the slots that were read, pair members swapped (kernel.cpp:1339-1345)
	bool all = true;
#pragma unroll
	for (int v = 0; v < V; v++) all = all && active[v];
	auto put = [&](raw* p, uint32_t xs, auto aligned, const raw* t) {
		if (all) store_raw<raw, V, decltype(aligned)::value>(p, xs, t);
		else {
#pragma unroll
			for (int v = 0; v < V; v++) if (active[v]) p[xs+(uint32_t)v] = t[v];
		}
	};
	{
		raw t[V];
#pragma unroll
		for (int v = 0; v < V; v++) t[v] = S::encode(f[v][0]);
		put(fi+row0, x, std::true_type{}, t);
	}
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I, cx = dir<Q>(i,0), cy = dir<Q>(i,1), cz = dir<Q>(i,2);
		const uint32_t sA = a.odd ? i : i+1, sB = a.odd ? i+1 : i;
		raw t[V];
#pragma unroll
		for (int v = 0; v < V; v++) t[v] = S::encode(f[v][i+1]);
		put(fi+(uint64_t)sA*a.N+row0, x, std::true_type{}, t);
#pragma unroll
		for (int v = 0; v < V; v++) t[v] = S::encode(f[v][i]);
		raw* pn = fi+(uint64_t)sB*a.N+((uint64_t)zq[cz == 0 ? 0 : cz > 0 ? 1 : 2]*a.Ny+yq[cy == 0 ? 0 : cy > 0 ? 1 : 2])*a.Nx;
		if constexpr (cx == 0) put(pn, x, std::true_type{}, t);
		else if constexpr (cx > 0) {
			if (x+(uint32_t)V < a.Nx) put(pn, x+1u, std::false_type{}, t);
			else {
#pragma unroll
				for (int v = 0; v < V; v++) if (active[v]) { const uint32_t xn = x+(uint32_t)v+1u; pn[xn == a.Nx ? 0u : xn] = t[v]; }
			}
		} else {
			if (x > 0u) put(pn, x-1u, std::false_type{}, t);
			else {
#pragma unroll
				for (int v = 0; v < V; v++) if (active[v]) { const uint32_t xv = x+(uint32_t)v; pn[xv == 0u ? a.Nx-1u : xv-1u] = t[v]; }
			}
		}
	});
}

#ifdef LBM_SMALL_KERNELS // the one-cell-per-thread kernels are compiled once, in lbmhip.hip
// ---------------------------------------------------------------------------------------------- one-cell-per-thread helpers
struct Cell { uint32_t x, y, z; uint64_t n; bool valid; };
__device__ __forceinline__ Cell cell_of_thread(const Args& a) { // grid = (ceil(Nx*Ny/block), Nz): one lattice plane per blockIdx.y
	Cell c;
	const uint64_t plane = (uint64_t)a.Nx*a.Ny, t = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
	c.valid = t < plane;
	c.z = blockIdx.y;
	c.x = (uint32_t)(t%a.Nx); c.y = (uint32_t)(t/a.Nx);
	c.n = (uint64_t)c.z*plane+t;
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
		f[i]   = S::decode(fi[(uint64_t)(odd ? i : i+1)*a.N+n]);
		f[i+1] = S::decode(fi[(uint64_t)(odd ? i+1 : i)*a.N+j[i]]);
	});
}

// FluidX3D-master/src/kernel.cpp:1364-1446 (without SURFACE/TEMPERATURE)
template <int Q, int ST> __global__ __launch_bounds__(256) void k_initialize(const Args a) {
	using S = Store<ST>; using raw = typename S::raw;
	const Cell c = cell_of_thread(a);
	if (!c.valid || is_halo(a, c)) return;
	const uint64_t n = c.n;
	uint64_t j[Q]; neighbors<Q>(a, c.x, c.y, c.z, j);
	uint8_t fl = a.flags[n];
	const uint8_t bo = fl&TYPE_BO;
	const bool moving_ext = (a.features&F_MOVING_BOUNDARIES) != 0u;
	if (bo == TYPE_S) {
		bool only_solid = true;
		static_for<1, Q, 1>([&](auto I) { only_solid = only_solid && (a.flags[j[I]]&TYPE_BO) == TYPE_S; });
		if (only_solid || !moving_ext) { a.u[n] = 0.0f; a.u[a.N+n] = 0.0f; a.u[2ull*a.N+n] = 0.0f; }
	} else if (moving_ext && bo != TYPE_E) {
		bool moving = false;
		static_for<1, Q, 1>([&](auto I) {
			const uint64_t ji = j[I];
			moving = moving || ((a.flags[ji]&TYPE_BO) == TYPE_S && (a.u[ji] != 0.0f || a.u[a.N+ji] != 0.0f || a.u[2ull*a.N+ji] != 0.0f));
		});
		fl = moving ? (uint8_t)(fl|TYPE_MS) : (uint8_t)(fl&~TYPE_MS);
		a.flags[n] = fl;
	}
	float feq[Q];
	f_eq<Q>(a.rho[n], a.u[n], a.u[a.N+n], a.u[2ull*a.N+n], feq);
	raw* fi = static_cast<raw*>(a.fi); // store_f with t=1 (kernel.cpp:1445)
	fi[n] = S::encode(feq[0]);
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I;
		fi[(uint64_t)(i+1)*a.N+j[i]] = S::encode(feq[i]);
		fi[(uint64_t)i*a.N+n] = S::encode(feq[i+1]);
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
						if (!h.insert) buf[k] = fi[(uint64_t)(a.odd ? opp : i)*a.N+((i%2) ? j[i] : n)];                 // kernel.cpp:2135
						else fi[(uint64_t)(a.odd ? i : opp)*a.N+((i%2) ? n : j[(i%2) ? i : i-1])] = buf[k];            // kernel.cpp:2144
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

#endif // LBM_SMALL_KERNELS

} // namespace lbm
