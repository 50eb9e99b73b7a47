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
	uint64_t S;          // distance between two DDF streams in elements (>= N)
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

// ---------------------------------------------------------------------------------------------- one cell or a pair of cells
// The cell arithmetic is written once over a type R: float (one cell) or float2v (two neighbouring cells of a lane).  With
// float2v the additions, multiplications and fmas become the packed FP32 instructions of gfx90a+ (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32: two IEEE results per issue slot); division, square root and min/max have no packed form and
// are evaluated element by element with the same scalar code as for one cell, so both types give the same bits.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <class R> struct lanes_of { static constexpr int W = 1; };
template <> struct lanes_of<float2v> { static constexpr int W = 2; };
__device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ float2v fma_(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float div_(float a, float b) { return a/b; }
__device__ __forceinline__ float2v div_(float2v a, float2v b) { float2v r; r.x = a.x/b.x; r.y = a.y/b.y; return r; }
__device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }
__device__ __forceinline__ float2v sqrt_(float2v a) { float2v r; r.x = sqrtf(a.x); r.y = sqrtf(a.y); return r; }
__device__ __forceinline__ float get_(float a, int) { return a; }
__device__ __forceinline__ float get_(float2v a, int k) { return k ? a.y : a.x; }
__device__ __forceinline__ void set_(float& a, int, float v) { a = v; }
__device__ __forceinline__ void set_(float2v& a, int k, float v) { if (k) a.y = v; else a.x = v; }
// m[k] ? a : b per cell
__device__ __forceinline__ float select_(const bool* m, float a, float b) { return m[0] ? a : b; }
__device__ __forceinline__ float2v select_(const bool* m, float2v a, float2v b) { float2v r; r.x = m[0] ? a.x : b.x; r.y = m[1] ? a.y : b.y; return r; }
template <class R> __device__ __forceinline__ R load_(const float* p) { return *reinterpret_cast<const R*>(p); } // p is aligned to sizeof(R)
// Memory access as wave-uniform base + 32-bit byte offset of the lane: the base stays in scalar registers and the access
// needs no 64-bit vector address arithmetic (global_load/store with an SGPR base).  Offsets are below 2^32 because a
// lattice row is (Nx < 2^30 cells, checked when a domain is created).
// Values travel as 1, 2, 4, 8 or 16 plain bytes (vectors of 16-bit elements as dwords).
template <int BYTES> struct bytes_of;
template <> struct bytes_of<1> { typedef uint8_t type; };
template <> struct bytes_of<2> { typedef uint16_t type; };
template <> struct bytes_of<4> { typedef uint32_t type; };
template <> struct bytes_of<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct bytes_of<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <class T, class B> __device__ __forceinline__ T ld(const B* uniform_base, const uint32_t byte_offset) {
	typedef typename bytes_of<sizeof(T)>::type W;
	return __builtin_bit_cast(T, *reinterpret_cast<const W*>(reinterpret_cast<const char*>(uniform_base)+byte_offset));
}
// The same offset as a value the optimiser cannot trace back: address arithmetic written with it is not merged with the
// arithmetic of an earlier program section (merged, the full 64-bit vector addresses would be carried across the section
// in registers instead of being formed by the store instruction from scalar base + offset).
__device__ __forceinline__ uint32_t fresh(uint32_t v) { asm("" : "+v"(v)); return v; }
template <class T, class B> __device__ __forceinline__ void st(B* uniform_base, const uint32_t byte_offset, const T value) {
	typedef typename bytes_of<sizeof(T)>::type W;
	*reinterpret_cast<W*>(reinterpret_cast<char*>(uniform_base)+byte_offset) = __builtin_bit_cast(W, value);
}

// The same accesses at agent scope, for data handed from one workgroup to another inside one launch (k_collide_pair): a
// store that is written through to memory instead of staying dirty in the L2 of the storing XCD, and a load that does not
// take its data from the CU's L1.  Elements of 1, 2 or 4 bytes.
enum : int { MEM_PLAIN = 0, MEM_STORE_THROUGH = 1, MEM_LOAD_COHERENT = 2 };
template <class T, int MEM, class B> __device__ __forceinline__ T ld_ddf(const B* uniform_base, const uint32_t byte_offset) {
	if constexpr (MEM == MEM_LOAD_COHERENT) {
		typedef typename bytes_of<sizeof(T)>::type W;
		static_assert(sizeof(T) <= 4, "agent-scope accesses move single elements");
		W* p = reinterpret_cast<W*>(const_cast<char*>(reinterpret_cast<const char*>(uniform_base))+byte_offset);
		return __builtin_bit_cast(T, __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	} else return ld<T>(uniform_base, byte_offset);
}
template <int MEM, class T, class B> __device__ __forceinline__ void st_ddf(B* uniform_base, const uint32_t byte_offset, const T value) {
	if constexpr (MEM == MEM_STORE_THROUGH) {
		typedef typename bytes_of<sizeof(T)>::type W;
		static_assert(sizeof(T) <= 4, "agent-scope accesses move single elements");
		__hip_atomic_store(reinterpret_cast<W*>(reinterpret_cast<char*>(uniform_base)+byte_offset), __builtin_bit_cast(W, value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	} else st(uniform_base, byte_offset, value);
}

// c_i . (a,b,c) with zero components skipped, accumulated in x,y,z order
template <int Q, int I, class R> __device__ __forceinline__ R cdot(R a, R b, R c) {
	constexpr int cx = dir<Q>(I,0), cy = dir<Q>(I,1), cz = dir<Q>(I,2);
	R s = R(0.0f);
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
	// two neighbouring cells at once
	typedef float2v raw2;
	static __device__ __forceinline__ float2v decode2(raw2 r) { return r; }
	static __device__ __forceinline__ raw2 encode2(float2v x) { return x; }
	static __device__ __forceinline__ raw2 merge2(raw2 old, raw2 val, bool keep0, bool keep1) { raw2 r; r.x = keep0 ? old.x : val.x; r.y = keep1 ? old.y : val.y; return r; }
	// encoding in two stages (see FP16C): nothing to prepare here
	template <int N, class R> static __device__ __forceinline__ void prescale(R*) {}
	static __device__ __forceinline__ raw encode_scaled(float x) { return encode(x); }
	static __device__ __forceinline__ raw2 encode2_scaled(float2v x) { return encode2(x); }
};
// FP16S: IEEE half scaled by 2^15 (lbm.cpp:572-575)
template <> struct Store<ST_FP16S> {
	using raw = uint16_t;
	static __device__ __forceinline__ float decode(raw r) { return __half2float(__ushort_as_half(r))*3.0517578E-5f; }
	// x*2^15 rounded to half once (v_fma_mixlo_f16; the scaling is exact, so this is the reference's multiply followed by
	// vstore_half_rte).  The addend is -0: the compiler's own fusion of the multiply and the conversion adds +0, which turns
	// a stored -0 into +0.
	static __device__ __forceinline__ raw encode(float x) {
		uint32_t r;
		asm("v_fma_mixlo_f16 %0, %1, %2, neg(0)" : "=v"(r) : "v"(x), "s"(32768.0f));
		return (raw)r;
	}
	// One of the two DDFs of a dword (h = 0: low half, 1: high half) without taking the dword apart.  decode_half is the
	// conversion and the scaling in one instruction (the product of a half and 2^-15 is exact before its one rounding, the
	// addend -0 leaves it as it is); encode_half is encode() writing its half of w in place and keeping the other half.
	static __device__ __forceinline__ float decode_half(uint32_t w, int h) {
		float r;
		if (h == 0) asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "s"(3.0517578E-5f));
		else asm("v_fma_mix_f32 %0, %1, %2, neg(0) op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "s"(3.0517578E-5f));
		return r;
	}
	static __device__ __forceinline__ void encode_half(uint32_t& w, int h, float x) {
		if (h == 0) asm("v_fma_mixlo_f16 %0, %1, %2, neg(0)" : "+v"(w) : "v"(x), "s"(32768.0f));
		else asm("v_fma_mixhi_f16 %0, %1, %2, neg(0)" : "+v"(w) : "v"(x), "s"(32768.0f));
	}
	// two neighbouring cells at once: one dword, the first cell in its low half; conversion with v_cvt_pk_f16_f32 (round to nearest even)
	typedef uint32_t raw2;
	static __device__ __forceinline__ float2v decode2(raw2 r) { return __builtin_convertvector(__builtin_bit_cast(half2v, r), float2v)*3.0517578E-5f; }
	static __device__ __forceinline__ raw2 encode2(float2v x) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(x*32768.0f, half2v)); }
	static __device__ __forceinline__ raw2 merge2(raw2 old, raw2 val, bool keep0, bool keep1) { const uint32_t m = (keep0 ? 0x0000FFFFu : 0u)|(keep1 ? 0xFFFF0000u : 0u); return (old&m)|(val&~m); }
	template <int N, class R> static __device__ __forceinline__ void prescale(R*) {}
	static __device__ __forceinline__ raw encode_scaled(float x) { return encode(x); }
	static __device__ __forceinline__ raw2 encode2_scaled(float2v x) { return encode2(x); }
};
// FP16C: 1 sign, 4 exponent (bias 15), 11 mantissa bits, subnormals below 2^-14 (kernel.cpp:858-869).
//
// Both directions go through FP32 values scaled by 2^-112, whose FP32 bit pattern IS the 16-bit code shifted left by 12:
// the code's exponent field (bias 15) sits in the low four bits of the FP32 exponent field (bias 127 = 15+112) and a code
// with exponent 0 (value mantissa * 2^-25) is an FP32 subnormal.  FP32 subnormals are kept (the default mode of HIP on
// gfx9), so one multiply normalises or denormalises exactly and no case distinction is left:
//   decode: move sign, exponent and mantissa into place, multiply by 2^112 (exact).
//   encode: multiply by 2^-112 ROUNDING TOWARDS ZERO, add half of the last kept bit (0x800) to the bit pattern, cut.
//           In the normal range the multiply is exact and this is the reference's add-and-truncate; below 2^-14 the
//           truncating multiply is the reference's right shift of the mantissa (kernel.cpp:868) and the add its
//           round-half-up.  Identical bits for every finite float (checked for all 2^32 inputs, see tests); only the
//           payload of a NaN differs.
// The rounding mode is a wave-wide register (MODE.fp_round), so the multiplies run inside asm blocks that set it to
// round-towards-zero and back; nothing the compiler schedules can land between the two s_setreg of a block.
#define LBM_RTZ_ON "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\t"
#define LBM_RTZ_OFF "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
#define LBM_RTZ_MUL(k) "v_mul_f32 %" #k ", 0x07800000, %" #k "\n\t"
#define LBM_RTZ_PKMUL(k, s) "v_pk_mul_f32 %" #k ", %" #k ", %" #s "\n\t"
// p[0..N) *= 2^-112, rounded towards zero; N <= 9 per asm block
template <int N> __device__ __forceinline__ void rtz_scale_block(float* p) {
	static_assert(N >= 1 && N <= 9, "block of 1..9 values");
	if constexpr (N == 1) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_OFF : "+v"(p[0]));
	else if constexpr (N == 2) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]));
	else if constexpr (N == 3) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]));
	else if constexpr (N == 4) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_MUL(3) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
	else if constexpr (N == 5) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_MUL(3) LBM_RTZ_MUL(4) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]));
	else if constexpr (N == 6) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_MUL(3) LBM_RTZ_MUL(4) LBM_RTZ_MUL(5) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]));
	else if constexpr (N == 7) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_MUL(3) LBM_RTZ_MUL(4) LBM_RTZ_MUL(5) LBM_RTZ_MUL(6) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]));
	else if constexpr (N == 8) asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_MUL(3) LBM_RTZ_MUL(4) LBM_RTZ_MUL(5) LBM_RTZ_MUL(6) LBM_RTZ_MUL(7) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
	else asm volatile(LBM_RTZ_ON LBM_RTZ_MUL(0) LBM_RTZ_MUL(1) LBM_RTZ_MUL(2) LBM_RTZ_MUL(3) LBM_RTZ_MUL(4) LBM_RTZ_MUL(5) LBM_RTZ_MUL(6) LBM_RTZ_MUL(7) LBM_RTZ_MUL(8) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]));
}
// the same for pairs of cells (one packed multiply per pair); s = {2^-112, 2^-112}
template <int N> __device__ __forceinline__ void rtz_scale_block(float2v* p) {
	static_assert(N >= 1 && N <= 9, "block of 1..9 pairs");
	const float2v s = {0x1p-112f, 0x1p-112f};
	if constexpr (N == 1) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 1) LBM_RTZ_OFF : "+v"(p[0]) : "v"(s));
	else if constexpr (N == 2) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 2) LBM_RTZ_PKMUL(1, 2) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]) : "v"(s));
	else if constexpr (N == 3) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 3) LBM_RTZ_PKMUL(1, 3) LBM_RTZ_PKMUL(2, 3) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]) : "v"(s));
	else if constexpr (N == 4) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 4) LBM_RTZ_PKMUL(1, 4) LBM_RTZ_PKMUL(2, 4) LBM_RTZ_PKMUL(3, 4) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(s));
	else if constexpr (N == 5) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 5) LBM_RTZ_PKMUL(1, 5) LBM_RTZ_PKMUL(2, 5) LBM_RTZ_PKMUL(3, 5) LBM_RTZ_PKMUL(4, 5) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]) : "v"(s));
	else if constexpr (N == 6) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 6) LBM_RTZ_PKMUL(1, 6) LBM_RTZ_PKMUL(2, 6) LBM_RTZ_PKMUL(3, 6) LBM_RTZ_PKMUL(4, 6) LBM_RTZ_PKMUL(5, 6) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]) : "v"(s));
	else if constexpr (N == 7) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 7) LBM_RTZ_PKMUL(1, 7) LBM_RTZ_PKMUL(2, 7) LBM_RTZ_PKMUL(3, 7) LBM_RTZ_PKMUL(4, 7) LBM_RTZ_PKMUL(5, 7) LBM_RTZ_PKMUL(6, 7) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]) : "v"(s));
	else if constexpr (N == 8) asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 8) LBM_RTZ_PKMUL(1, 8) LBM_RTZ_PKMUL(2, 8) LBM_RTZ_PKMUL(3, 8) LBM_RTZ_PKMUL(4, 8) LBM_RTZ_PKMUL(5, 8) LBM_RTZ_PKMUL(6, 8) LBM_RTZ_PKMUL(7, 8) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(s));
	else asm volatile(LBM_RTZ_ON LBM_RTZ_PKMUL(0, 9) LBM_RTZ_PKMUL(1, 9) LBM_RTZ_PKMUL(2, 9) LBM_RTZ_PKMUL(3, 9) LBM_RTZ_PKMUL(4, 9) LBM_RTZ_PKMUL(5, 9) LBM_RTZ_PKMUL(6, 9) LBM_RTZ_PKMUL(7, 9) LBM_RTZ_PKMUL(8, 9) LBM_RTZ_OFF : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]) : "v"(s));
}
template <int N, class R> __device__ __forceinline__ void rtz_scale(R* p) { // all N values, nine per block
	static_for<0, N/9, 1>([&](auto B) { rtz_scale_block<9>(p+9*(int)B); });
	if constexpr (N%9 != 0) rtz_scale_block<N%9>(p+N/9*9);
}
template <> struct Store<ST_FP16C> {
	using raw = uint16_t;
	static __device__ __forceinline__ float decode(raw r) {
		const int32_t t = (int32_t)((uint32_t)r<<16)>>4; // sign | sign copies | exponent | mantissa
		return __uint_as_float((uint32_t)t&0x87FFF000u)*0x1p112f;
	}
	// the 16-bit code of a value that was already multiplied by 2^-112 towards zero (rtz_scale)
	static __device__ __forceinline__ raw encode_scaled(float w) {
		const uint32_t u = __float_as_uint(w)+0x00000800u;
		return (raw)(((u>>12)&0x7FFFu)|((u>>16)&0x8000u));
	}
	static __device__ __forceinline__ raw encode(float x) { rtz_scale_block<1>(&x); return encode_scaled(x); }
	template <int N> static __device__ __forceinline__ void prescale(float* f) { rtz_scale<N>(f); }
	template <int N> static __device__ __forceinline__ void prescale(float2v* f) { rtz_scale<N>(f); }
	// One of the two codes of a dword (h = 0: low half, 1: high half) without taking the dword apart.  decode_half: the low
	// half arrives sign-extended and shifted in one instruction (sub-dword operand select); encode_half (value already scaled
	// by 2^-112 towards zero): m holds the code in its bits 27..12 - exponent and mantissa where they are, the sign moved from
	// bit 31 to bit 27 - and one shift with a sub-dword destination writes it into its half of w, keeping the other half.
	static __device__ __forceinline__ float decode_half(uint32_t w, int h) {
		uint32_t t;
		if (h == 0) asm("v_lshlrev_b32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(t) : "s"(12u), "v"(w));
		else t = (uint32_t)((int32_t)w>>4);
		return __uint_as_float(t&0x87FFF000u)*0x1p112f;
	}
	static __device__ __forceinline__ void encode_half(uint32_t& w, int h, float scaled) {
		const uint32_t u = __float_as_uint(scaled)+0x00000800u;
		uint32_t m; // (u & mask) | (u>>4 & ~mask) in one instruction
		asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(m) : "s"(0x07FFF000u), "v"(u), "v"(u>>4));
		if (h == 0) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "s"(12u), "v"(m));
		else asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "s"(12u), "v"(m));
	}
	typedef uint32_t raw2;
	static __device__ __forceinline__ float2v decode2(raw2 r) {
		float2v v;
		v.x = __uint_as_float((uint32_t)((int32_t)(r<<16)>>4)&0x87FFF000u);
		v.y = __uint_as_float((uint32_t)((int32_t)r>>4)&0x87FFF000u);
		return v*0x1p112f;
	}
	static __device__ __forceinline__ raw2 encode2_scaled(float2v w) {
		const uint32_t u0 = __float_as_uint(w.x)+0x00000800u, u1 = __float_as_uint(w.y)+0x00000800u;
		const uint32_t lo = ((u0>>12)&0x7FFFu)|((u0>>16)&0x8000u);
		const uint32_t hi = ((u1<<4)&0x7FFF0000u)|(u1&0x80000000u);
		return hi|lo;
	}
	static __device__ __forceinline__ raw2 encode2(float2v x) { rtz_scale_block<1>(&x); return encode2_scaled(x); }
	static __device__ __forceinline__ raw2 merge2(raw2 old, raw2 val, bool keep0, bool keep1) { const uint32_t m = (keep0 ? 0x0000FFFFu : 0u)|(keep1 ? 0xFFFF0000u : 0u); return (old&m)|(val&~m); }
};

// ---------------------------------------------------------------------------------------------- moments, equilibrium, forcing
// kernel.cpp:1010-1067 (shifted equilibrium, pair-symmetric nested fma).  EqTerms holds what all directions share; the
// equilibrium of one opposite pair is a separate function so that a collision without subgrid model can form it where it
// is used (27 + 27 values live at the same time otherwise, which costs a wave of occupancy with 16-bit D3Q27 lattices).
template <class R> struct EqTerms { R rho, rhom1, c3, ux3, uy3, uz3; };
template <int Q, class R> __device__ __forceinline__ EqTerms<R> eq_terms(R rho, R ux, R uy, R uz) {
	EqTerms<R> e;
	e.rho = rho; e.rhom1 = rho-1.0f;
	if constexpr (Q == 9) e.c3 = -3.0f*(ux*ux+uy*uy); else e.c3 = -3.0f*(ux*ux+uy*uy+uz*uz);
	e.ux3 = ux*3.0f; e.uy3 = uy*3.0f; e.uz3 = uz*3.0f;
	return e;
}
template <int Q, class R> __device__ __forceinline__ R f_eq_rest(const EqTerms<R>& e) { return weight<Q>(0)*fma_(e.rho, 0.5f*e.c3, e.rhom1); }
template <int Q, int I, class R> __device__ __forceinline__ void f_eq_pair(const EqTerms<R>& e, R& feq_i, R& feq_j) { // directions I and I+1
	const R cu = cdot<Q, I>(e.ux3, e.uy3, e.uz3);
	const R wr = weight<Q>(I)*e.rho, wm = weight<Q>(I)*e.rhom1;
	feq_i = fma_(wr, fma_(R(0.5f), fma_(cu, cu, e.c3), cu), wm);
	feq_j = fma_(wr, fma_(R(0.5f), fma_(cu, cu, e.c3), -cu), wm);
}
template <int Q, class R> __device__ __forceinline__ void f_eq(R rho, R ux, R uy, R uz, R* feq) {
	const EqTerms<R> e = eq_terms<Q>(rho, ux, uy, uz);
	feq[0] = f_eq_rest<Q>(e);
	static_for<1, Q, 2>([&](auto I) { f_eq_pair<Q, I>(e, feq[I], feq[I+1]); });
}
// kernel.cpp:1069-1094 (1 added last; momentum as alternating +/- chain over the pairs in ascending order).  The four
// sums are independent chains; they are written pair by pair, side by side, so that consecutive instructions do not
// depend on each other (each chain keeps its own order of additions, which is what fixes the bits).
template <int Q, class R> __device__ __forceinline__ void rho_u(const R* f, R& rhon, R& uxn, R& uyn, R& uzn) {
	R rho = f[0];
	R m[3] = {R(0.0f), R(0.0f), R(0.0f)};
	bool started[3] = {false, false, false};
	static_for<1, Q, 2>([&](auto I) {
		constexpr int i = I;
		rho += f[i];
		static_for<0, 3, 1>([&](auto K) {
			constexpr int k = K, ck = dir<Q>(i, k);
			if constexpr (ck != 0) {
				const R plus = ck > 0 ? f[i] : f[i+1], minus = ck > 0 ? f[i+1] : f[i];
				if (!started[k]) { m[k] = plus-minus; started[k] = true; } else { m[k] = m[k]+plus; m[k] = m[k]-minus; }
			}
		});
		rho += f[i+1];
	});
	rho += 1.0f;
	rhon = rho; uxn = div_(m[0], rho); uyn = div_(m[1], rho); uzn = div_(m[2], rho);
}
__device__ __forceinline__ float clamp_c(float x) { return fminf(fmaxf(x, -0.57735027f), 0.57735027f); } // lbm.cpp:507
__device__ __forceinline__ float2v clamp_c(float2v x) { float2v r; r.x = clamp_c(x.x); r.y = clamp_c(x.y); return r; }

// Guo forcing term of direction I (kernel.cpp:1097-1107)
template <int Q, int I, class R> __device__ __forceinline__ R forcing_term(R ux, R uy, R uz, R fx, R fy, R fz, R uF) {
	if constexpr (I == 0) return 9.0f*weight<Q>(0)*uF;
	else return 9.0f*weight<Q>(I)*fma_(cdot<Q, I>(fx, fy, fz), cdot<Q, I>(ux, uy, uz)+0.33333334f, uF);
}

// feature-mask functors: compile-time constant or read from the launch arguments
template <uint32_t MASK> struct StaticFeat { static constexpr uint32_t mask = MASK; __device__ __forceinline__ constexpr bool operator()(uint32_t bit) const { return (MASK&bit) != 0u; } };
struct DynamicFeat { uint32_t mask; __device__ __forceinline__ bool operator()(uint32_t bit) const { return (mask&bit) != 0u; } };
template <class Feat> struct fixed_features { static constexpr bool value = false; static constexpr uint32_t mask = 0u; };
template <uint32_t MASK> struct fixed_features<StaticFeat<MASK>> { static constexpr bool value = true; static constexpr uint32_t mask = MASK; };

// ---------------------------------------------------------------------------------------------- the collision of one cell (R = float) or of cells n, n+1 (R = float2v)
// f[] holds the streamed-in DDFs; on return it holds the post-collision DDFs (kernel.cpp:1497-1651).
// FEAT is the feature mask (compile-time when the kernel is specialised, run-time otherwise); bo[] and write_fields[] have one entry per cell.
template <int Q, int COLL, bool FIELDS_ONLY, class Feat, class R>
__device__ __forceinline__ void collide(R* f, const Feat feat, const uint8_t* bo, const Args& a, const uint64_t row0, const uint32_t x,
                                        R& rhon, R& uxn, R& uyn, R& uzn, bool* write_fields) {
	constexpr int W = lanes_of<R>::W;
	const uint64_t n = row0+x; // the cell (the first of the pair); row0 is wave-uniform
	bool eq_cell[W];
#pragma unroll
	for (int k = 0; k < W; k++) eq_cell[k] = feat(F_EQUILIBRIUM_BOUNDARIES) && bo[k] == TYPE_E;
	if (W == 1 && !FIELDS_ONLY && eq_cell[0]) { rhon = R(a.rho[n]); uxn = R(a.u[n]); uyn = R(a.u[a.N+n]); uzn = R(a.u[2ull*a.N+n]); }
	else {
		rho_u<Q>(f, rhon, uxn, uyn, uzn);
		if constexpr (W > 1 && !FIELDS_ONLY) {
#pragma unroll
			for (int k = 0; k < W; k++) if (eq_cell[k]) { set_(rhon, k, a.rho[n+k]); set_(uxn, k, a.u[n+k]); set_(uyn, k, a.u[a.N+n+k]); set_(uzn, k, a.u[2ull*a.N+n+k]); }
		}
	}
	R fxn = R(a.fx), fyn = R(a.fy), fzn = R(a.fz);
	if (feat(F_FORCE_FIELD)) { fxn += ld<R>(a.F+row0, 4u*x); fyn += ld<R>(a.F+a.N+row0, 4u*x); fzn += ld<R>(a.F+2ull*a.N+row0, 4u*x); }
	if (feat(F_VOLUME_FORCE)) {
		const R rho2 = div_(R(0.5f), rhon);
		uxn = clamp_c(fma_(fxn, rho2, uxn)); uyn = clamp_c(fma_(fyn, rho2, uyn)); uzn = clamp_c(fma_(fzn, rho2, uzn));
	} else {
		uxn = clamp_c(uxn); uyn = clamp_c(uyn); uzn = clamp_c(uzn);
	}
#pragma unroll
	for (int k = 0; k < W; k++) write_fields[k] = (FIELDS_ONLY || feat(F_UPDATE_FIELDS)) && !eq_cell[k];
	if constexpr (FIELDS_ONLY) return;

	const bool force = feat(F_VOLUME_FORCE);
	R uF = R(0.0f);
	if (force) { if constexpr (Q == 9) uF = -0.33333334f*fma_(uxn, fxn, uyn*fyn); else uF = -0.33333334f*fma_(uxn, fxn, fma_(uyn, fyn, uzn*fzn)); }
	const bool any_eq = feat(F_EQUILIBRIUM_BOUNDARIES);

	// With a fixed relaxation rate (no subgrid model, known at compile time) every pair's equilibrium is formed where the
	// collision uses it; the subgrid model needs all of them first (its rate depends on the whole non-equilibrium part).
	constexpr bool lazy = fixed_features<Feat>::value && (fixed_features<Feat>::mask&F_SUBGRID) == 0u;
	const EqTerms<R> e = eq_terms<Q>(rhon, uxn, uyn, uzn);
	R feq[lazy ? 1 : Q];
	if constexpr (!lazy) {
		feq[0] = f_eq_rest<Q>(e);
		static_for<1, Q, 2>([&](auto I) { f_eq_pair<Q, I>(e, feq[I], feq[I+1]); });
	}
	auto eq_rest = [&]() -> R { if constexpr (lazy) return f_eq_rest<Q>(e); else return feq[0]; };
	auto eq_pair = [&](auto I, R& ei, R& ej) { if constexpr (lazy) f_eq_pair<Q, I>(e, ei, ej); else { ei = feq[I]; ej = feq[I+1]; } };
	R w = R(a.w);
	if (feat(F_SUBGRID)) { // Smagorinsky-Lilly (kernel.cpp:1597-1611)
		if constexpr (!lazy) {
			const float tau0 = a.tau0; // = 1/w (kernel.cpp:1598)
			R Hxx = R(0.0f), Hyy = R(0.0f), Hzz = R(0.0f), Hxy = R(0.0f), Hxz = R(0.0f), Hyz = R(0.0f);
			static_for<1, Q, 1>([&](auto I) {
				constexpr int i = I, cx = dir<Q>(i,0), cy = dir<Q>(i,1), cz = dir<Q>(i,2);
				const R fneq = f[i]-feq[i];
				if constexpr (cx != 0) Hxx += fneq;
				if constexpr (cx*cy != 0) Hxy += cx*cy > 0 ? fneq : -fneq;
				if constexpr (cy != 0) Hyy += fneq;
				if constexpr (cx*cz != 0) Hxz += cx*cz > 0 ? fneq : -fneq;
				if constexpr (cy*cz != 0) Hyz += cy*cz > 0 ? fneq : -fneq;
				if constexpr (cz != 0) Hzz += fneq;
			});
			const R Qn = Hxx*Hxx+Hyy*Hyy+Hzz*Hzz+2.0f*(Hxy*Hxy+Hxz*Hxz+Hyz*Hyz);
			w = div_(R(2.0f), tau0+sqrt_(tau0*tau0+div_(0.76421222f*sqrt_(Qn), rhon)));
		}
	}
	if constexpr (COLL == COLL_SRT) { // kernel.cpp:1613-1622
		const R c_tau = fma_(w, R(-0.5f), R(1.0f)), omw = 1.0f-w;
		auto relax = [&](auto I, const R eq) {
			constexpr int i = I;
			const R Fin = force ? forcing_term<Q, i>(uxn, uyn, uzn, fxn, fyn, fzn, uF)*c_tau : R(0.0f);
			const R fc = fma_(omw, f[i], fma_(w, eq, Fin));
			f[i] = any_eq ? select_(eq_cell, eq, fc) : fc;
		};
		relax(std::integral_constant<int, 0>{}, eq_rest());
		static_for<1, Q, 2>([&](auto I) {
			R ei, ej;
			eq_pair(I, ei, ej);
			relax(I, ei);
			relax(std::integral_constant<int, I+1>{}, ej);
		});
	} else { // TRT (kernel.cpp:1623-1651)
		const R wp = w, wm = feat(F_SUBGRID) ? div_(R(1.0f), div_(R(0.1875f), div_(R(1.0f), w)-0.5f)+0.5f) : R(a.wm); // without subgrid w is the lattice constant
		const R c_taup = fma_(wp, R(-0.25f), R(0.5f)), c_taum = fma_(wm, R(-0.25f), R(0.5f)), hwp = 0.5f*wp, hwm = 0.5f*wm;
		{
			const R e0 = eq_rest();
			const R Fin0 = force ? forcing_term<Q, 0>(uxn, uyn, uzn, fxn, fyn, fzn, uF) : R(0.0f);
			const R F0 = force ? fma_(c_taup, Fin0+Fin0, c_taum*(Fin0-Fin0)) : R(0.0f);
			const R fc = fma_(hwp, e0-f[0]+e0-f[0], fma_(hwm, e0-e0-f[0]+f[0], f[0]+F0));
			f[0] = any_eq ? select_(eq_cell, e0, fc) : fc;
		}
		static_for<1, Q, 2>([&](auto I) {
			constexpr int i = I, j = I+1;
			R Fi = R(0.0f), Fj = R(0.0f);
			if (force) {
				const R Gi = forcing_term<Q, i>(uxn, uyn, uzn, fxn, fyn, fzn, uF), Gj = forcing_term<Q, j>(uxn, uyn, uzn, fxn, fyn, fzn, uF);
				Fi = fma_(c_taup, Gi+Gj, c_taum*(Gi-Gj));
				Fj = fma_(c_taup, Gj+Gi, c_taum*(Gj-Gi));
			}
			R ei, ej;
			eq_pair(I, ei, ej);
			const R fi_ = f[i], fj_ = f[j];
			// ci = hwp*(ei-fi+ej-fj) + (hwm*(ei-ej-fi+fj) + (fi+Fi)), cj likewise with i and j swapped (kernel.cpp:1640-1649); the four
			// sums are written side by side, one addition of each at a time, so that neighbouring instructions are independent
			R pi = ei-fi_, mi = ei-ej, pj = ej-fj_, mj = ej-ei;
			pi = pi+ej; mi = mi-fi_; pj = pj+ei; mj = mj-fj_;
			pi = pi-fj_; mi = mi+fj_; pj = pj-fi_; mj = mj+fi_;
			const R ci = fma_(hwp, pi, fma_(hwm, mi, fi_+Fi));
			const R cj = fma_(hwp, pj, fma_(hwm, mj, fj_+Fj));
			f[i] = any_eq ? select_(eq_cell, ei, ci) : ci;
			f[j] = any_eq ? select_(eq_cell, ej, cj) : cj;
		});
	}
}

} // namespace lbm
