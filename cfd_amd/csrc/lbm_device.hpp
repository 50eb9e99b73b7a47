// lbm_device.hpp - device-side building blocks of the lattice-Boltzmann step for gfx950 (wave64).
//
// Everything here is a template over the velocity set Q, the DDF storage format ST and compile-time
// feature masks, resolved with static_for so that direction vectors, weights and slot indices are
// literal constants in the generated code.  Arithmetic is FP32 in the operation order of the reference's
// kernels (FluidX3D-master/src/kernel.cpp:1010-1118, 1332-1345, 1597-1651) so that FP32 results are
// reproducible bit for bit against the CPU oracle; build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace lbm {

enum : int { ST_FP32 = 0, ST_FP16S = 1, ST_FP16C = 2 };
enum : int { COLL_SRT = 0, COLL_TRT = 1 };
enum : uint32_t { F_UPDATE_FIELDS = 1u, F_VOLUME_FORCE = 2u, F_MOVING_BOUNDARIES = 4u, F_EQUILIBRIUM_BOUNDARIES = 8u, F_FORCE_FIELD = 16u, F_SUBGRID = 32u };
enum : uint8_t { TYPE_S = 0x01, TYPE_E = 0x02, TYPE_MS = 0x03, TYPE_BO = 0x03, TYPE_T = 0x04 }; // lbm.cpp:537-551

// arguments shared by all lattice kernels
struct Args {
	void* fi; float* rho; float* u; uint8_t* flags; float* F;
	uint64_t N;          // cells in the local lattice
	uint32_t Nx, Ny, Nz; // local lattice (with halo)
	uint32_t hx, hy, hz; // 1 where the axis has a halo layer on both ends
	uint32_t z0, z1;     // planes handled by this launch
	uint32_t odd;        // t & 1
	uint32_t features;   // run-time feature mask (used by kernels instantiated with FEAT < 0)
	uint32_t xchunks;    // thread blocks per lattice row
	float fx, fy, fz, w;
	float tau0, wm;      // 1/w and the TRT rate 1/(0.1875/(1/w-0.5)+0.5) of the lattice viscosity, evaluated once on the host in float
};

// ---------------------------------------------------------------------------------------------- static_for
template <int Begin, int End, int Step, class Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
	if constexpr (Begin < End) {
		fn(std::integral_constant<int, Begin>{});
		static_for<Begin+Step, End, Step>(fn);
	}
}

// ---------------------------------------------------------------------------------------------- velocity sets
// direction vectors in the reference's order (kernel.cpp:874-895): 0 rest, then opposite pairs (odd i, i+1)
template <int Q> __host__ __device__ constexpr int dir(int i, int axis) {
	if constexpr (Q == 9) {
		constexpr int t[9][3] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{1,1,0},{-1,-1,0},{1,-1,0},{-1,1,0}};
		return t[i][axis];
	} else if constexpr (Q == 15) {
		constexpr int t[15][3] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1},{1,1,1},{-1,-1,-1},{1,1,-1},{-1,-1,1},{1,-1,1},{-1,1,-1},{-1,1,1},{1,-1,-1}};
		return t[i][axis];
	} else {
		constexpr int t[27][3] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1},
			{1,1,0},{-1,-1,0},{1,0,1},{-1,0,-1},{0,1,1},{0,-1,-1},{1,-1,0},{-1,1,0},{1,0,-1},{-1,0,1},{0,1,-1},{0,-1,1},
			{1,1,1},{-1,-1,-1},{1,1,-1},{-1,-1,1},{1,-1,1},{-1,1,-1},{-1,1,1},{1,-1,-1}};
		return t[i][axis];
	}
}
// lattice weights by |c|^2 (lbm.cpp:512-529), evaluated in float like the reference's macros
template <int Q> __host__ __device__ constexpr float weight(int i) {
	const int k = dir<Q>(i,0)*dir<Q>(i,0)+dir<Q>(i,1)*dir<Q>(i,1)+dir<Q>(i,2)*dir<Q>(i,2);
	if constexpr (Q == 9) return k == 0 ? 1.0f/2.25f : k == 1 ? 1.0f/9.0f : 1.0f/36.0f;
	else if constexpr (Q == 15) return k == 0 ? 1.0f/4.5f : k == 1 ? 1.0f/9.0f : 1.0f/72.0f;
	else if constexpr (Q == 19) return k == 0 ? 1.0f/3.0f : k == 1 ? 1.0f/18.0f : 1.0f/36.0f;
	else return k == 0 ? 1.0f/3.375f : k == 1 ? 1.0f/13.5f : k == 2 ? 1.0f/54.0f : 1.0f/216.0f;
}
template <int Q> __host__ __device__ constexpr int transfers() { return Q == 9 ? 3 : Q == 27 ? 9 : 5; } // lbm.cpp:5-21
// DDFs crossing the + face of an axis, in the reference's buffer order (kernel.cpp:2098-2129); the - face uses the opposite of each
template <int Q> __host__ __device__ constexpr int transfer_dir(int axis, int b) {
	if constexpr (Q == 9) { constexpr int t[2][3] = {{1,5,7},{3,5,8}}; return t[axis][b]; }
	else if constexpr (Q == 15) { constexpr int t[3][5] = {{1,7,14,9,11},{3,7,12,9,13},{5,7,10,11,13}}; return t[axis][b]; }
	else if constexpr (Q == 19) { constexpr int t[3][5] = {{1,7,13,9,15},{3,7,14,11,17},{5,9,16,11,18}}; return t[axis][b]; }
	else { constexpr int t[3][9] = {{1,7,13,9,15,19,26,21,23},{3,7,14,11,17,19,24,21,25},{5,9,16,11,18,19,22,23,25}}; return t[axis][b]; }
}

// c_i . (a,b,c) with zero components skipped, accumulated in x,y,z order
template <int Q, int I> __device__ __forceinline__ float cdot(float a, float b, float c) {
	constexpr int cx = dir<Q>(I,0), cy = dir<Q>(I,1), cz = dir<Q>(I,2);
	float s = 0.0f;
	if constexpr (cx != 0) s = cx > 0 ? a : -a;
	if constexpr (cy != 0) { if constexpr (cx == 0) s = cy > 0 ? b : -b; else s = cy > 0 ? s+b : s-b; }
	if constexpr (cz != 0) { if constexpr (cx == 0 && cy == 0) s = cz > 0 ? c : -c; else s = cz > 0 ? s+c : s-c; }
	return s;
}

// ---------------------------------------------------------------------------------------------- DDF storage
template <int ST> struct Store;
template <> struct Store<ST_FP32> {
	using raw = float;
	static __device__ __forceinline__ float decode(raw r) { return r; }
	static __device__ __forceinline__ raw encode(float x) { return x; }
};
// FP16S: IEEE half scaled by 2^15 (lbm.cpp:572-575)
template <> struct Store<ST_FP16S> {
	using raw = uint16_t;
	static __device__ __forceinline__ float decode(raw r) { return __half2float(__ushort_as_half(r))*3.0517578E-5f; }
	static __device__ __forceinline__ raw encode(float x) { return __half_as_ushort(__float2half_rn(x*32768.0f)); }
};
// FP16C: 1 sign, 4 exponent (bias 15), 11 mantissa bits, subnormals below 2^-14 (kernel.cpp:858-869)
template <> struct Store<ST_FP16C> {
	using raw = uint16_t;
	// exponent and mantissa moved into the FP32 fields give x*2^-112 (an FP32 subnormal when e=0); the multiply by
	// 2^112 is exact and normalises both cases, so no branch is needed (FP32 subnormals are kept in the default mode)
	static __device__ __forceinline__ float decode(raw r) {
		const uint32_t h = r;
		const float v = __uint_as_float((h&0x7FFFu)<<12)*0x1p112f;
		return __uint_as_float(__float_as_uint(v)|((h&0x8000u)<<16));
	}
	// round by adding half of the last kept mantissa bit and truncating; subnormal results restore the implicit one first
	static __device__ __forceinline__ raw encode(float x) {
		const uint32_t b = __float_as_uint(x)+0x00000800u, a = b&0x7FFFFFFFu, e = a>>23;
		const uint32_t normal = ((a>>12)-0x38000u)&0x7FFFu;
		const uint32_t subnormal = ((((0x007FF800u+(a&0x007FFFFFu))>>((124u-e)&31u))+1u)>>1);
		return (raw)(((b>>16)&0x8000u)|(e > 112u ? normal : e > 100u ? subnormal : 0u));
	}
};

// ---------------------------------------------------------------------------------------------- moments, equilibrium, forcing
// kernel.cpp:1010-1067 (shifted equilibrium, pair-symmetric nested fma)
template <int Q> __device__ __forceinline__ void f_eq(float rho, float ux, float uy, float uz, float* feq) {
	const float rhom1 = rho-1.0f;
	float c3;
	if constexpr (Q == 9) c3 = -3.0f*(ux*ux+uy*uy); else c3 = -3.0f*(ux*ux+uy*uy+uz*uz);
	ux *= 3.0f; uy *= 3.0f; uz *= 3.0f;
	feq[0] = weight<Q>(0)*fmaf(rho, 0.5f*c3, rhom1);
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I;
		const float cu = cdot<Q, i>(ux, uy, uz);
		const float wr = weight<Q>(i)*rho, wm = weight<Q>(i)*rhom1;
		feq[i]   = fmaf(wr, fmaf(0.5f, fmaf(cu, cu, c3), cu), wm);
		feq[i+1] = fmaf(wr, fmaf(0.5f, fmaf(cu, cu, c3), -cu), wm);
	});
}
// kernel.cpp:1069-1094 (1 added last; momentum as alternating +/- chain over the pairs in ascending order)
template <int Q> __device__ __forceinline__ void rho_u(const float* f, float& rhon, float& uxn, float& uyn, float& uzn) {
	float rho = f[0];
	static_for<1, Q, 1>([&](auto I) { rho += f[I]; });
	rho += 1.0f;
	float m[3] = {0.0f, 0.0f, 0.0f};
	static_for<0, 3, 1>([&](auto K) {
		constexpr int k = K;
		bool started = false;
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I, ck = dir<Q>(i, k);
			if constexpr (ck != 0) {
				const float plus = ck > 0 ? f[i] : f[i+1], minus = ck > 0 ? f[i+1] : f[i];
				if (!started) { m[k] = plus-minus; started = true; } else { m[k] = m[k]+plus; m[k] = m[k]-minus; }
			}
		});
	});
	rhon = rho; uxn = m[0]/rho; uyn = m[1]/rho; uzn = m[2]/rho;
}
__device__ __forceinline__ float clamp_c(float x) { return fminf(fmaxf(x, -0.57735027f), 0.57735027f); } // lbm.cpp:507

// Guo forcing term of direction I (kernel.cpp:1097-1107)
template <int Q, int I> __device__ __forceinline__ float forcing_term(float ux, float uy, float uz, float fx, float fy, float fz, float uF) {
	if constexpr (I == 0) return 9.0f*weight<Q>(0)*uF;
	else return 9.0f*weight<Q>(I)*fmaf(cdot<Q, I>(fx, fy, fz), cdot<Q, I>(ux, uy, uz)+0.33333334f, uF);
}

// ---------------------------------------------------------------------------------------------- the collision of one cell
// f[] holds the streamed-in DDFs; on return it holds the post-collision DDFs (kernel.cpp:1497-1651).
// FEAT is the feature mask (compile-time when the kernel is specialised, run-time otherwise).
template <int Q, int COLL, bool FIELDS_ONLY, class Feat>
__device__ __forceinline__ void collide(float* f, const Feat feat, const uint8_t bo, const Args& a, const uint64_t n,
                                        float& rhon, float& uxn, float& uyn, float& uzn, bool& write_fields) {
	const bool eq_cell = feat(F_EQUILIBRIUM_BOUNDARIES) && bo == TYPE_E;
	if (!FIELDS_ONLY && eq_cell) { rhon = a.rho[n]; uxn = a.u[n]; uyn = a.u[a.N+n]; uzn = a.u[2ull*a.N+n]; }
	else rho_u<Q>(f, rhon, uxn, uyn, uzn);
	float fxn = a.fx, fyn = a.fy, fzn = a.fz;
	if (feat(F_FORCE_FIELD)) { fxn += a.F[n]; fyn += a.F[a.N+n]; fzn += a.F[2ull*a.N+n]; }
	if (feat(F_VOLUME_FORCE)) {
		const float rho2 = 0.5f/rhon;
		uxn = clamp_c(fmaf(fxn, rho2, uxn)); uyn = clamp_c(fmaf(fyn, rho2, uyn)); uzn = clamp_c(fmaf(fzn, rho2, uzn));
	} else {
		uxn = clamp_c(uxn); uyn = clamp_c(uyn); uzn = clamp_c(uzn);
	}
	write_fields = (FIELDS_ONLY || feat(F_UPDATE_FIELDS)) && !eq_cell;
	if constexpr (FIELDS_ONLY) return;

	float feq[Q];
	f_eq<Q>(rhon, uxn, uyn, uzn, feq);
	float w = a.w;
	if (feat(F_SUBGRID)) { // Smagorinsky-Lilly (kernel.cpp:1597-1611)
		const float tau0 = a.tau0; // = 1/w (kernel.cpp:1598)
		float Hxx = 0.0f, Hyy = 0.0f, Hzz = 0.0f, Hxy = 0.0f, Hxz = 0.0f, Hyz = 0.0f;
		static_for<1, Q, 1>([&](auto I) {
			constexpr int i = I, cx = dir<Q>(i,0), cy = dir<Q>(i,1), cz = dir<Q>(i,2);
			const float fneq = f[i]-feq[i];
			if constexpr (cx != 0) Hxx += fneq;
			if constexpr (cx*cy != 0) Hxy += cx*cy > 0 ? fneq : -fneq;
			if constexpr (cy != 0) Hyy += fneq;
			if constexpr (cx*cz != 0) Hxz += cx*cz > 0 ? fneq : -fneq;
			if constexpr (cy*cz != 0) Hyz += cy*cz > 0 ? fneq : -fneq;
			if constexpr (cz != 0) Hzz += fneq;
		});
		const float Qn = Hxx*Hxx+Hyy*Hyy+Hzz*Hzz+2.0f*(Hxy*Hxy+Hxz*Hxz+Hyz*Hyz);
		w = 2.0f/(tau0+sqrtf(tau0*tau0+0.76421222f*sqrtf(Qn)/rhon));
	}
	const bool force = feat(F_VOLUME_FORCE);
	float uF = 0.0f;
	if (force) { if constexpr (Q == 9) uF = -0.33333334f*fmaf(uxn, fxn, uyn*fyn); else uF = -0.33333334f*fmaf(uxn, fxn, fmaf(uyn, fyn, uzn*fzn)); }
	if constexpr (COLL == COLL_SRT) { // kernel.cpp:1613-1622
		const float c_tau = fmaf(w, -0.5f, 1.0f), omw = 1.0f-w;
		static_for<0, Q, 1>([&](auto I) {
			constexpr int i = I;
			const float Fin = force ? forcing_term<Q, i>(uxn, uyn, uzn, fxn, fyn, fzn, uF)*c_tau : 0.0f;
			f[i] = eq_cell ? feq[i] : fmaf(omw, f[i], fmaf(w, feq[i], Fin));
		});
	} else { // TRT (kernel.cpp:1623-1651)
		const float wp = w, wm = feat(F_SUBGRID) ? 1.0f/(0.1875f/(1.0f/w-0.5f)+0.5f) : a.wm; // without subgrid w is the lattice constant
		const float c_taup = fmaf(wp, -0.25f, 0.5f), c_taum = fmaf(wm, -0.25f, 0.5f), hwp = 0.5f*wp, hwm = 0.5f*wm;
		{
			const float Fin0 = force ? forcing_term<Q, 0>(uxn, uyn, uzn, fxn, fyn, fzn, uF) : 0.0f;
			const float F0 = force ? fmaf(c_taup, Fin0+Fin0, c_taum*(Fin0-Fin0)) : 0.0f;
			f[0] = eq_cell ? feq[0] : fmaf(hwp, feq[0]-f[0]+feq[0]-f[0], fmaf(hwm, feq[0]-feq[0]-f[0]+f[0], f[0]+F0));
		}
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I, j = I+1;
			float Fi = 0.0f, Fj = 0.0f;
			if (force) {
				const float Gi = forcing_term<Q, i>(uxn, uyn, uzn, fxn, fyn, fzn, uF), Gj = forcing_term<Q, j>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
				Fi = fmaf(c_taup, Gi+Gj, c_taum*(Gi-Gj));
				Fj = fmaf(c_taup, Gj+Gi, c_taum*(Gj-Gi));
			}
			const float fi_ = f[i], fj_ = f[j], ei = feq[i], ej = feq[j];
			f[i] = eq_cell ? ei : fmaf(hwp, ei-fi_+ej-fj_, fmaf(hwm, ei-ej-fi_+fj_, fi_+Fi));
			f[j] = eq_cell ? ej : fmaf(hwp, ej-fj_+ei-fi_, fmaf(hwm, ej-ei-fj_+fi_, fj_+Fj));
		});
	}
}

// feature-mask functors: compile-time constant or read from the launch arguments
template <uint32_t MASK> struct StaticFeat { __device__ __forceinline__ constexpr bool operator()(uint32_t bit) const { return (MASK&bit) != 0u; } };
struct DynamicFeat { uint32_t mask; __device__ __forceinline__ bool operator()(uint32_t bit) const { return (mask&bit) != 0u; } };

} // namespace lbm
