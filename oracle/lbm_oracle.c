/*
 * lbm_oracle.c - CPU restatement of the FluidX3D lattice-Boltzmann time step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (cfd_amd/, include/) never does.  It is a scalar, table-driven restatement written from the algorithm
 * of the reference's OpenCL kernels; every function names the reference lines it follows
 * (paths relative to /root/reference/FluidX3D-master/src/).
 *
 * Parity pin: the reference ships no golden vectors or result-pinning tests, so this restatement is
 * pinned against the reference's own kernels compiled for the host CPU (oracle/ref/build_ref.py ->
 * oracle/_ref/libref_*.so) and against fixtures generated from them (tests/golden/, made by
 * tests/golden/make_golden.py).
 *
 * Data layout (kernel.cpp:827-833, 871-873, 970-972): cell n = x+(y+z*Ny)*Nx; DDFs SoA fi[i*N+n] stored
 * shifted by the lattice weight (f_i - w_i); u SoA u[n], u[N+n], u[2N+n]; rho float; flags uchar.
 * Everything is evaluated in float with explicit fmaf() where the reference writes fma(); build with
 * -ffp-contract=off so nothing else is fused.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { ST_FP32 = 0, ST_FP16S = 1, ST_FP16C = 2 };
enum { COLL_SRT = 0, COLL_TRT = 1 };
enum { /* feature bits, same numbering as include/lbmhip.h */
	FEAT_UPDATE_FIELDS = 1, FEAT_VOLUME_FORCE = 2, FEAT_MOVING_BOUNDARIES = 4,
	FEAT_EQUILIBRIUM_BOUNDARIES = 8, FEAT_FORCE_FIELD = 16, FEAT_SUBGRID = 32
};
enum { TYPE_S = 1, TYPE_E = 2, TYPE_MS = 3, TYPE_BO = 3, TYPE_T = 4 }; /* lbm.cpp:537-551 */

typedef struct {
	uint32_t Nx, Ny, Nz; /* local grid including halo layers */
	uint32_t Dx, Dy, Dz; /* domains per axis; an axis with D>1 carries a 1-cell halo on both sides */
	uint32_t Q;          /* 9, 15, 19 or 27 */
	uint32_t storage, collision, features;
	float w;             /* relaxation rate 1/(3 nu + 0.5), lbm.cpp:509-511 */
} oracle_cfg;

/* ------------------------------------------------------------------------------------------------
 * velocity sets: direction vectors in the reference's order (kernel.cpp:874-895); odd i and i+1 are
 * opposite.  Weights by |c|^2 class (lbm.cpp:512-529).  Halo transfer lists (kernel.cpp:2098-2129).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int8_t x, y, z; } dir3;
static const dir3 C9[9] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{1,1,0},{-1,-1,0},{1,-1,0},{-1,1,0}};
static const dir3 C15[15] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1},{1,1,1},{-1,-1,-1},
	{1,1,-1},{-1,-1,1},{1,-1,1},{-1,1,-1},{-1,1,1},{1,-1,-1}};
static const dir3 C27[27] = {{0,0,0},{1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1},
	{1,1,0},{-1,-1,0},{1,0,1},{-1,0,-1},{0,1,1},{0,-1,-1},{1,-1,0},{-1,1,0},{1,0,-1},{-1,0,1},{0,1,-1},{0,-1,1},
	{1,1,1},{-1,-1,-1},{1,1,-1},{-1,-1,1},{1,-1,1},{-1,1,-1},{-1,1,1},{1,-1,-1}}; /* D3Q19 = first 19 entries */

static const dir3* dirs(uint32_t Q) { return Q == 9u ? C9 : Q == 15u ? C15 : C27; }
static uint32_t dimensions(uint32_t Q) { return Q == 9u ? 2u : 3u; }
static uint32_t transfers(uint32_t Q) { return Q == 9u ? 3u : Q == 27u ? 9u : 5u; } /* lbm.cpp:5-21 */

static void weights(uint32_t Q, float* w) {
	float w0, ws, we = 0.0f, wc = 0.0f; /* centre, straight, edge, corner */
	if (Q == 9u) { w0 = 1.0f/2.25f; ws = 1.0f/9.0f; we = 1.0f/36.0f; }
	else if (Q == 15u) { w0 = 1.0f/4.5f; ws = 1.0f/9.0f; wc = 1.0f/72.0f; }
	else if (Q == 19u) { w0 = 1.0f/3.0f; ws = 1.0f/18.0f; we = 1.0f/36.0f; }
	else { w0 = 1.0f/3.375f; ws = 1.0f/13.5f; we = 1.0f/54.0f; wc = 1.0f/216.0f; }
	const dir3* c = dirs(Q);
	for (uint32_t i = 0u; i < Q; i++) {
		const int k = c[i].x*c[i].x + c[i].y*c[i].y + c[i].z*c[i].z;
		w[i] = k == 0 ? w0 : k == 1 ? ws : k == 2 ? we : wc;
	}
}

/* the DDFs that cross a domain face: for side 2*axis+0 (plus face) every direction with c_axis=+1, in
 * the reference's buffer order; side 2*axis+1 uses the opposite direction (i+1) of each */
static uint32_t transfer_direction(uint32_t Q, uint32_t side, uint32_t b) {
	static const uint8_t T9[2][3] = {{1,5,7},{3,5,8}};
	static const uint8_t T15[3][5] = {{1,7,14,9,11},{3,7,12,9,13},{5,7,10,11,13}};
	static const uint8_t T19[3][5] = {{1,7,13,9,15},{3,7,14,11,17},{5,9,16,11,18}};
	static const uint8_t T27[3][9] = {{1,7,13,9,15,19,26,21,23},{3,7,14,11,17,19,24,21,25},{5,9,16,11,18,19,22,23,25}};
	const uint32_t axis = side/2u;
	const uint32_t p = Q == 9u ? T9[axis][b] : Q == 15u ? T15[axis][b] : Q == 19u ? T19[axis][b] : T27[axis][b];
	if (side%2u == 0u) return p;
	return p%2u ? p+1u : p-1u; /* opposite direction */
}

/* ------------------------------------------------------------------------------------------------
 * DDF storage formats (lbm.cpp:561-609; kernel.cpp:858-869)
 * ---------------------------------------------------------------------------------------------- */
static uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

static float half_to_float(uint16_t h) { /* IEEE-754 binary16 -> binary32, exact */
	const uint32_t s = (uint32_t)(h&0x8000u)<<16, e = (h>>10)&0x1Fu, m = h&0x3FFu;
	if (e == 0u) {
		if (m == 0u) return u2f(s);
		const float v = (float)m*5.9604644775390625e-8f; /* m * 2^-24 */
		return (h&0x8000u) ? -v : v;
	}
	if (e == 31u) return u2f(s|0x7F800000u|(m<<13));
	return u2f(s|((e+112u)<<23)|(m<<13));
}
static uint16_t float_to_half_rte(float x) { /* binary32 -> binary16, round to nearest even */
	const uint32_t b = f2u(x), s = (b>>16)&0x8000u, a = b&0x7FFFFFFFu;
	if (a >= 0x7F800000u) return (uint16_t)(s|0x7C00u|(a > 0x7F800000u ? 0x200u : 0u));
	if (a >= 0x477FF000u) return (uint16_t)(s|0x7C00u); /* rounds to >= 65520 -> inf */
	if (a < 0x33000001u) return (uint16_t)s;            /* <= 2^-25 -> 0 (tie goes to even = 0) */
	const int32_t e = (int32_t)(a>>23)-127;
	uint32_t m = (a&0x007FFFFFu)|0x00800000u;
	uint32_t shift, he;
	if (e < -14) { shift = (uint32_t)(13-14-e); he = 0u; } else { shift = 13u; he = (uint32_t)(e+15); }
	const uint32_t q = m>>shift, rem = m&((1u<<shift)-1u), halfway = 1u<<(shift-1u);
	uint32_t r = (he == 0u ? q : (he<<10)+(q-0x400u));
	if (rem > halfway || (rem == halfway && (q&1u))) r++; /* carries propagate into the exponent correctly */
	return (uint16_t)(s|r);
}

/* FP16C: 1 sign, 4 exponent (bias 15), 11 mantissa bits; subnormals below 2^-14 (kernel.cpp:858-869) */
static float fp16c_to_float(uint16_t h) {
	const uint32_t s = (uint32_t)(h&0x8000u)<<16, e = (h>>11)&0xFu, m = h&0x7FFu;
	if (e != 0u) return u2f(s|((e+112u)<<23)|(m<<12));
	if (m == 0u) return u2f(s);
	uint32_t lead = 10u; /* position of the leading one of the 11-bit mantissa */
	while (!((m>>lead)&1u)) lead--;
	const uint32_t frac = (m<<(23u-lead))&0x007FFFFFu; /* leading one shifted out to the implicit bit */
	return u2f(s|((lead+102u)<<23)|(frac&0x007FF000u)); /* value = m * 2^-25 */
}
static uint16_t float_to_fp16c(float x) {
	const uint32_t b = f2u(x)+0x00000800u; /* add half of the last kept mantissa bit, then truncate */
	const uint32_t s = (b&0x80000000u)>>16, e = (b&0x7F800000u)>>23, m = b&0x007FFFFFu;
	if (e > 112u) return (uint16_t)(s|(((e-112u)<<11)&0x7800u)|(m>>12)); /* normal range, |x| assumed < 2 */
	if (e > 100u) { /* subnormal: restore the implicit one, undo the pre-rounding, shift, round half up */
		const uint32_t full = 0x00800000u-0x00000800u+m;
		return (uint16_t)(s|(((full>>(124u-e))+1u)>>1));
	}
	return (uint16_t)s;
}

static float ddf_load(const oracle_cfg* cfg, const void* fi, uint64_t k) {
	if (cfg->storage == ST_FP32) return ((const float*)fi)[k];
	if (cfg->storage == ST_FP16S) return half_to_float(((const uint16_t*)fi)[k])*3.0517578E-5f;
	return fp16c_to_float(((const uint16_t*)fi)[k]);
}
static void ddf_store(const oracle_cfg* cfg, void* fi, uint64_t k, float x) {
	if (cfg->storage == ST_FP32) ((float*)fi)[k] = x;
	else if (cfg->storage == ST_FP16S) ((uint16_t*)fi)[k] = float_to_half_rte(x*32768.0f);
	else ((uint16_t*)fi)[k] = float_to_fp16c(x);
}

/* ------------------------------------------------------------------------------------------------
 * indexing (kernel.cpp:827-833, 850-853, 914-968)
 * ---------------------------------------------------------------------------------------------- */
static uint64_t cells(const oracle_cfg* cfg) { return (uint64_t)cfg->Nx*cfg->Ny*cfg->Nz; }
static void coordinates(const oracle_cfg* cfg, uint64_t n, uint32_t* x, uint32_t* y, uint32_t* z) {
	const uint64_t plane = (uint64_t)cfg->Nx*cfg->Ny, t = n%plane;
	*x = (uint32_t)(t%cfg->Nx); *y = (uint32_t)(t/cfg->Nx); *z = (uint32_t)(n/plane);
}
static int is_halo(const oracle_cfg* cfg, uint32_t x, uint32_t y, uint32_t z) {
	return (cfg->Dx > 1u && (x == 0u || x >= cfg->Nx-1u)) || (cfg->Dy > 1u && (y == 0u || y >= cfg->Ny-1u)) ||
	       (cfg->Dz > 1u && (z == 0u || z >= cfg->Nz-1u));
}
/* neighbour of (x,y,z) in direction i with periodic wrap on the local grid */
static void neighbors(const oracle_cfg* cfg, uint32_t x, uint32_t y, uint32_t z, uint64_t* j) {
	const dir3* c = dirs(cfg->Q);
	for (uint32_t i = 0u; i < cfg->Q; i++) {
		const uint32_t xn = (x+cfg->Nx+(uint32_t)(int32_t)c[i].x)%cfg->Nx;
		const uint32_t yn = (y+cfg->Ny+(uint32_t)(int32_t)c[i].y)%cfg->Ny;
		const uint32_t zn = (z+cfg->Nz+(uint32_t)(int32_t)c[i].z)%cfg->Nz;
		j[i] = (uint64_t)xn+((uint64_t)yn+(uint64_t)zn*cfg->Ny)*cfg->Nx;
	}
}

/* ------------------------------------------------------------------------------------------------
 * esoteric-pull streaming (kernel.cpp:1332-1345): every cell reads and writes the same 2 slots per
 * direction pair; the slot index swaps with the parity of t
 * ---------------------------------------------------------------------------------------------- */
static void load_f(const oracle_cfg* cfg, const void* fi, uint64_t n, const uint64_t* j, uint64_t t, float* f) {
	const uint64_t N = cells(cfg);
	const uint32_t odd = (uint32_t)(t&1u);
	f[0] = ddf_load(cfg, fi, n);
	for (uint32_t i = 1u; i < cfg->Q; i += 2u) {
		f[i]    = ddf_load(cfg, fi, (uint64_t)(odd ? i : i+1u)*N+n);
		f[i+1u] = ddf_load(cfg, fi, (uint64_t)(odd ? i+1u : i)*N+j[i]);
	}
}
static void store_f(const oracle_cfg* cfg, void* fi, uint64_t n, const uint64_t* j, uint64_t t, const float* f) {
	const uint64_t N = cells(cfg);
	const uint32_t odd = (uint32_t)(t&1u);
	ddf_store(cfg, fi, n, f[0]);
	for (uint32_t i = 1u; i < cfg->Q; i += 2u) {
		ddf_store(cfg, fi, (uint64_t)(odd ? i+1u : i)*N+j[i], f[i]);
		ddf_store(cfg, fi, (uint64_t)(odd ? i : i+1u)*N+n, f[i+1u]);
	}
}

/* ------------------------------------------------------------------------------------------------
 * moments and equilibrium (kernel.cpp:1010-1094)
 * ---------------------------------------------------------------------------------------------- */
static float signed_sum(const dir3 c, float a, float b, float d) { /* c.x*a + c.y*b + c.z*d, zero terms skipped, x,y,z order */
	float s = 0.0f; int started = 0;
	const int8_t cc[3] = {c.x, c.y, c.z}; const float v[3] = {a, b, d};
	for (int k = 0; k < 3; k++) if (cc[k]) {
		if (!started) { s = cc[k] > 0 ? v[k] : -v[k]; started = 1; }
		else s = cc[k] > 0 ? s+v[k] : s-v[k];
	}
	return s;
}
static void f_eq(const oracle_cfg* cfg, float rho, float ux, float uy, float uz, float* feq) {
	const dir3* c = dirs(cfg->Q);
	float w[27]; weights(cfg->Q, w);
	const float rhom1 = rho-1.0f;
	const float c3 = cfg->Q == 9u ? -3.0f*(ux*ux+uy*uy) : -3.0f*(ux*ux+uy*uy+uz*uz);
	ux *= 3.0f; uy *= 3.0f; uz *= 3.0f;
	feq[0] = w[0]*fmaf(rho, 0.5f*c3, rhom1);
	for (uint32_t i = 1u; i < cfg->Q; i += 2u) {
		const float cu = signed_sum(c[i], ux, uy, uz);
		const float wr = w[i]*rho, wm = w[i]*rhom1;
		feq[i]    = fmaf(wr, fmaf(0.5f, fmaf(cu, cu, c3), cu), wm);
		feq[i+1u] = fmaf(wr, fmaf(0.5f, fmaf(cu, cu, c3), -cu), wm);
	}
}
static void rho_u(const oracle_cfg* cfg, const float* f, float* rhon, float* uxn, float* uyn, float* uzn) {
	const dir3* c = dirs(cfg->Q);
	float rho = f[0];
	for (uint32_t i = 1u; i < cfg->Q; i++) rho += f[i];
	rho += 1.0f; /* the shift is added last */
	float m[3] = {0.0f, 0.0f, 0.0f};
	for (int k = 0; k < 3; k++) { /* alternating + and -, pairs in ascending order, positive member first */
		int started = 0;
		for (uint32_t i = 1u; i < cfg->Q; i += 2u) {
			const int8_t ck = k == 0 ? c[i].x : k == 1 ? c[i].y : c[i].z;
			if (!ck) continue;
			const float plus = ck > 0 ? f[i] : f[i+1u], minus = ck > 0 ? f[i+1u] : f[i];
			if (!started) { m[k] = plus-minus; started = 1; } else { m[k] = m[k]+plus; m[k] = m[k]-minus; }
		}
	}
	*rhon = rho; *uxn = m[0]/rho; *uyn = m[1]/rho; *uzn = m[2]/rho;
}
static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#define LATTICE_C 0.57735027f /* lbm.cpp:507 */

/* Guo forcing terms (kernel.cpp:1097-1107) */
static void forcing_terms(const oracle_cfg* cfg, float ux, float uy, float uz, float fx, float fy, float fz, float* Fin) {
	const dir3* c = dirs(cfg->Q);
	float w[27]; weights(cfg->Q, w);
	const float uF = cfg->Q == 9u ? -0.33333334f*fmaf(ux, fx, uy*fy) : -0.33333334f*fmaf(ux, fx, fmaf(uy, fy, uz*fz));
	Fin[0] = 9.0f*w[0]*uF;
	for (uint32_t i = 1u; i < cfg->Q; i++) {
		const float cf = signed_sum(c[i], fx, fy, fz), cu = signed_sum(c[i], ux, uy, uz);
		Fin[i] = 9.0f*w[i]*fmaf(cf, cu+0.33333334f, uF);
	}
}

/* Dirichlet velocity on cells next to moving solids (kernel.cpp:1111-1118) */
static void moving_boundaries(const oracle_cfg* cfg, float* f, const uint64_t* j, const float* u, const uint8_t* flags) {
	const dir3* c = dirs(cfg->Q);
	float w[27]; weights(cfg->Q, w);
	const uint64_t N = cells(cfg);
	for (uint32_t i = 1u; i < cfg->Q; i += 2u) {
		const float w6 = -6.0f*w[i];
		uint64_t ji = j[i+1u];
		if ((flags[ji]&TYPE_BO) == TYPE_S) f[i] = fmaf(w6, signed_sum(c[i+1u], u[ji], u[N+ji], u[2u*N+ji]), f[i]);
		ji = j[i];
		if ((flags[ji]&TYPE_BO) == TYPE_S) f[i+1u] = fmaf(w6, signed_sum(c[i], u[ji], u[N+ji], u[2u*N+ji]), f[i+1u]);
	}
}

/* ------------------------------------------------------------------------------------------------
 * kernels
 * ---------------------------------------------------------------------------------------------- */
/* kernel.cpp:1364-1446 (SURFACE and TEMPERATURE parts are out of scope) */
void oracle_initialize(const oracle_cfg* cfg, void* fi, const float* rho, float* u, uint8_t* flags) {
	const uint64_t N = cells(cfg);
	for (uint64_t n = 0u; n < N; n++) {
		uint32_t x, y, z; coordinates(cfg, n, &x, &y, &z);
		if (is_halo(cfg, x, y, z)) continue;
		uint64_t j[27]; neighbors(cfg, x, y, z, j);
		uint8_t fl = flags[n];
		const uint8_t bo = fl&TYPE_BO;
		if (bo == TYPE_S) {
			int only_solid = 1;
			for (uint32_t i = 1u; i < cfg->Q; i++) only_solid = only_solid && (flags[j[i]]&TYPE_BO) == TYPE_S;
			if (only_solid || !(cfg->features&FEAT_MOVING_BOUNDARIES)) { u[n] = 0.0f; u[N+n] = 0.0f; u[2u*N+n] = 0.0f; }
		} else if ((cfg->features&FEAT_MOVING_BOUNDARIES) && bo != TYPE_E) {
			int moving = 0;
			for (uint32_t i = 1u; i < cfg->Q; i++) {
				const uint64_t ji = j[i];
				moving = moving || ((flags[ji]&TYPE_BO) == TYPE_S && (u[ji] != 0.0f || u[N+ji] != 0.0f || u[2u*N+ji] != 0.0f));
			}
			fl = moving ? (uint8_t)(fl|TYPE_MS) : (uint8_t)(fl&~TYPE_MS);
			flags[n] = fl;
		}
		float feq[27];
		f_eq(cfg, rho[n], u[n], u[N+n], u[2u*N+n], feq);
		store_f(cfg, fi, n, j, 1u, feq);
	}
}

/* shared front half of stream_collide / update_fields (kernel.cpp:1487-1578, 1826-1879) */
static void collide_cell(const oracle_cfg* cfg, void* fi, float* rho, float* u, const uint8_t* flags, uint64_t n,
                         uint32_t x, uint32_t y, uint32_t z, uint64_t t, float fx, float fy, float fz, const float* F, int fields_only) {
	const uint64_t N = cells(cfg);
	const uint32_t Q = cfg->Q, feat = cfg->features;
	const uint8_t bo = flags[n]&TYPE_BO;
	uint64_t j[27]; neighbors(cfg, x, y, z, j);
	float f[27]; load_f(cfg, fi, n, j, t, f);
	if ((feat&FEAT_MOVING_BOUNDARIES) && bo == TYPE_MS) moving_boundaries(cfg, f, j, u, flags);
	float rhon, uxn, uyn, uzn;
	if (!fields_only && (feat&FEAT_EQUILIBRIUM_BOUNDARIES) && bo == TYPE_E) { rhon = rho[n]; uxn = u[n]; uyn = u[N+n]; uzn = u[2u*N+n]; }
	else rho_u(cfg, f, &rhon, &uxn, &uyn, &uzn);
	float fxn = fx, fyn = fy, fzn = fz;
	if (feat&FEAT_FORCE_FIELD) { fxn += F[n]; fyn += F[N+n]; fzn += F[2u*N+n]; }
	float Fin[27];
	if (feat&FEAT_VOLUME_FORCE) {
		const float rho2 = 0.5f/rhon;
		uxn = clampf(fmaf(fxn, rho2, uxn), -LATTICE_C, LATTICE_C);
		uyn = clampf(fmaf(fyn, rho2, uyn), -LATTICE_C, LATTICE_C);
		uzn = clampf(fmaf(fzn, rho2, uzn), -LATTICE_C, LATTICE_C);
		if (!fields_only) forcing_terms(cfg, uxn, uyn, uzn, fxn, fyn, fzn, Fin);
	} else {
		uxn = clampf(uxn, -LATTICE_C, LATTICE_C); uyn = clampf(uyn, -LATTICE_C, LATTICE_C); uzn = clampf(uzn, -LATTICE_C, LATTICE_C);
		for (uint32_t i = 0u; i < Q; i++) Fin[i] = 0.0f;
	}
	if (fields_only || (feat&FEAT_UPDATE_FIELDS)) { /* kernel.cpp:1581-1591, 1881-1893 */
		if (!((feat&FEAT_EQUILIBRIUM_BOUNDARIES) && bo == TYPE_E)) { rho[n] = rhon; u[n] = uxn; u[N+n] = uyn; u[2u*N+n] = uzn; }
	}
	if (fields_only) return;

	float feq[27]; f_eq(cfg, rhon, uxn, uyn, uzn, feq);
	float w = cfg->w;
	if (feat&FEAT_SUBGRID) { /* Smagorinsky-Lilly (kernel.cpp:1597-1611) */
		const dir3* c = dirs(Q);
		const float tau0 = 1.0f/w;
		float Hxx = 0.0f, Hyy = 0.0f, Hzz = 0.0f, Hxy = 0.0f, Hxz = 0.0f, Hyz = 0.0f;
		for (uint32_t i = 1u; i < Q; i++) {
			const float fneq = f[i]-feq[i];
			const int cx = c[i].x, cy = c[i].y, cz = c[i].z;
			if (cx != 0) Hxx += fneq;
			if (cx*cy != 0) Hxy += cx*cy > 0 ? fneq : -fneq;
			if (cy != 0) Hyy += fneq;
			if (cx*cz != 0) Hxz += cx*cz > 0 ? fneq : -fneq;
			if (cy*cz != 0) Hyz += cy*cz > 0 ? fneq : -fneq;
			if (cz != 0) Hzz += fneq;
		}
		const float Qn = Hxx*Hxx+Hyy*Hyy+Hzz*Hzz+2.0f*(Hxy*Hxy+Hxz*Hxz+Hyz*Hyz);
		w = 2.0f/(tau0+sqrtf(tau0*tau0+0.76421222f*sqrtf(Qn)/rhon));
	}
	const int eq_cell = (feat&FEAT_EQUILIBRIUM_BOUNDARIES) && bo == TYPE_E;
	if (cfg->collision == COLL_SRT) { /* kernel.cpp:1613-1622 */
		if (feat&FEAT_VOLUME_FORCE) { const float c_tau = fmaf(w, -0.5f, 1.0f); for (uint32_t i = 0u; i < Q; i++) Fin[i] *= c_tau; }
		for (uint32_t i = 0u; i < Q; i++) f[i] = eq_cell ? feq[i] : fmaf(1.0f-w, f[i], fmaf(w, feq[i], Fin[i]));
	} else { /* TRT (kernel.cpp:1623-1651) */
		const float wp = w, wm = 1.0f/(0.1875f/(1.0f/w-0.5f)+0.5f);
		float bar[27];
		if (feat&FEAT_VOLUME_FORCE) {
			const float c_taup = fmaf(wp, -0.25f, 0.5f), c_taum = fmaf(wm, -0.25f, 0.5f);
			bar[0] = Fin[0];
			for (uint32_t i = 1u; i < Q; i += 2u) { bar[i] = Fin[i+1u]; bar[i+1u] = Fin[i]; }
			for (uint32_t i = 0u; i < Q; i++) Fin[i] = fmaf(c_taup, Fin[i]+bar[i], c_taum*(Fin[i]-bar[i]));
		}
		float fb[27], eb[27];
		fb[0] = f[0]; eb[0] = feq[0];
		for (uint32_t i = 1u; i < Q; i += 2u) { fb[i] = f[i+1u]; fb[i+1u] = f[i]; eb[i] = feq[i+1u]; eb[i+1u] = feq[i]; }
		for (uint32_t i = 0u; i < Q; i++)
			f[i] = eq_cell ? feq[i] : fmaf(0.5f*wp, feq[i]-f[i]+eb[i]-fb[i], fmaf(0.5f*wm, feq[i]-eb[i]-f[i]+fb[i], f[i]+Fin[i]));
	}
	store_f(cfg, fi, n, j, t, f);
}

/* kernel.cpp:1470-1654 */
void oracle_stream_collide(const oracle_cfg* cfg, void* fi, float* rho, float* u, const uint8_t* flags, uint64_t t,
                           float fx, float fy, float fz, const float* F) {
	const uint64_t N = cells(cfg);
	for (uint64_t n = 0u; n < N; n++) {
		uint32_t x, y, z; coordinates(cfg, n, &x, &y, &z);
		if (is_halo(cfg, x, y, z) || (flags[n]&TYPE_BO) == TYPE_S) continue;
		collide_cell(cfg, fi, rho, u, flags, n, x, y, z, t, fx, fy, fz, F, 0);
	}
}
/* kernel.cpp:1812-1894 */
void oracle_update_fields(const oracle_cfg* cfg, const void* fi, float* rho, float* u, const uint8_t* flags, uint64_t t,
                          float fx, float fy, float fz, const float* F) {
	const uint64_t N = cells(cfg);
	for (uint64_t n = 0u; n < N; n++) {
		uint32_t x, y, z; coordinates(cfg, n, &x, &y, &z);
		if (is_halo(cfg, x, y, z) || (flags[n]&TYPE_BO) == TYPE_S) continue;
		collide_cell(cfg, (void*)fi, rho, u, flags, n, x, y, z, t, fx, fy, fz, F, 1);
	}
}
/* kernel.cpp:1897-1917: the DDFs parked in a solid cell carry the momentum the fluid handed to it */
void oracle_update_force_field(const oracle_cfg* cfg, const void* fi, const uint8_t* flags, uint64_t t, float* F) {
	const uint64_t N = cells(cfg);
	for (uint64_t n = 0u; n < N; n++) {
		uint32_t x, y, z; coordinates(cfg, n, &x, &y, &z);
		if (is_halo(cfg, x, y, z) || (flags[n]&TYPE_BO) != TYPE_S) continue;
		uint64_t j[27]; neighbors(cfg, x, y, z, j);
		float f[27]; load_f(cfg, fi, n, j, t, f);
		float r, ax, ay, az; rho_u(cfg, f, &r, &ax, &ay, &az);
		F[n] = 2.0f*ax*r; F[N+n] = 2.0f*ay*r; F[2u*N+n] = 2.0f*az*r;
	}
}
void oracle_reset_force_field(const oracle_cfg* cfg, float* F) { memset(F, 0, (size_t)(3u*cells(cfg))*sizeof(float)); }

/* kernel.cpp:1449-1465 */
void oracle_update_moving_boundaries(const oracle_cfg* cfg, const float* u, uint8_t* flags) {
	const uint64_t N = cells(cfg);
	for (uint64_t n = 0u; n < N; n++) {
		uint32_t x, y, z; coordinates(cfg, n, &x, &y, &z);
		if (is_halo(cfg, x, y, z)) continue;
		const uint8_t fl = flags[n], bo = fl&TYPE_BO;
		if (bo == TYPE_S || bo == TYPE_E || (fl&TYPE_T)) continue;
		uint64_t j[27]; neighbors(cfg, x, y, z, j);
		int moving = 0;
		for (uint32_t i = 1u; i < cfg->Q; i++) {
			const uint64_t ji = j[i];
			moving = moving || ((u[ji] != 0.0f || u[N+ji] != 0.0f || u[2u*N+ji] != 0.0f) && (flags[ji]&TYPE_BO) == TYPE_S);
		}
		flags[n] = moving ? (uint8_t)(fl|TYPE_MS) : (uint8_t)(fl&~TYPE_MS);
	}
}

/* ------------------------------------------------------------------------------------------------
 * halo transfer (kernel.cpp:2077-2223).  Face cell a of an axis: extract reads the outermost interior
 * layer (N-2 on the plus side, 1 on the minus side), insert writes the halo layer (N-1 / 0).
 * Buffer layout buf[b*A+a].
 * ---------------------------------------------------------------------------------------------- */
static uint64_t face_area(const oracle_cfg* cfg, uint32_t axis) {
	return axis == 0u ? (uint64_t)cfg->Ny*cfg->Nz : axis == 1u ? (uint64_t)cfg->Nz*cfg->Nx : (uint64_t)cfg->Nx*cfg->Ny;
}
static void face_cell(const oracle_cfg* cfg, uint32_t axis, uint32_t a, uint32_t layer, uint32_t* x, uint32_t* y, uint32_t* z) {
	if (axis == 0u) { *x = layer; *y = a%cfg->Ny; *z = a/cfg->Ny; }
	else if (axis == 1u) { *x = a/cfg->Nz; *y = layer; *z = a%cfg->Nz; }
	else { *x = a%cfg->Nx; *y = a/cfg->Nx; *z = layer; }
}
static uint64_t cell_index(const oracle_cfg* cfg, uint32_t x, uint32_t y, uint32_t z) { return (uint64_t)x+((uint64_t)y+(uint64_t)z*cfg->Ny)*cfg->Nx; }

static void copy_raw(const oracle_cfg* cfg, void* dst, uint64_t kd, const void* src, uint64_t ks) {
	if (cfg->storage == ST_FP32) ((uint32_t*)dst)[kd] = ((const uint32_t*)src)[ks];
	else ((uint16_t*)dst)[kd] = ((const uint16_t*)src)[ks];
}
void oracle_transfer_fi(const oracle_cfg* cfg, int insert, uint32_t axis, uint64_t t, void* buf_p, void* buf_m, void* fi) {
	const uint64_t N = cells(cfg), A = face_area(cfg, axis);
	const uint32_t Nax = axis == 0u ? cfg->Nx : axis == 1u ? cfg->Ny : cfg->Nz, T = transfers(cfg->Q), odd = (uint32_t)(t&1u);
	for (uint32_t side = 0u; side < 2u; side++) {
		void* buf = side == 0u ? buf_p : buf_m;
		const uint32_t layer = insert ? (side == 0u ? Nax-1u : 0u) : (side == 0u ? Nax-2u : 1u);
		for (uint32_t a = 0u; a < (uint32_t)A; a++) {
			uint32_t x, y, z; face_cell(cfg, axis, a, layer, &x, &y, &z);
			const uint64_t n = cell_index(cfg, x, y, z);
			uint64_t j[27]; neighbors(cfg, x, y, z, j);
			for (uint32_t b = 0u; b < T; b++) {
				const uint32_t i = transfer_direction(cfg->Q, 2u*axis+side, b), opposite = i%2u ? i+1u : i-1u;
				if (!insert) { /* kernel.cpp:2135 */
					const uint64_t k = (uint64_t)(odd ? opposite : i)*N+(i%2u ? j[i] : n);
					copy_raw(cfg, buf, (uint64_t)b*A+a, fi, k);
				} else {       /* kernel.cpp:2144 */
					const uint64_t k = (uint64_t)(odd ? i : opposite)*N+(i%2u ? n : j[i-1u]);
					copy_raw(cfg, fi, k, buf, (uint64_t)b*A+a);
				}
			}
		}
	}
}
/* kernel.cpp:2161-2186: 4 floats (rho, ux, uy, uz) then 1 flag byte per face cell */
void oracle_transfer_rho_u_flags(const oracle_cfg* cfg, int insert, uint32_t axis, void* buf_p, void* buf_m, float* rho, float* u, uint8_t* flags) {
	const uint64_t N = cells(cfg), A = face_area(cfg, axis);
	const uint32_t Nax = axis == 0u ? cfg->Nx : axis == 1u ? cfg->Ny : cfg->Nz;
	for (uint32_t side = 0u; side < 2u; side++) {
		float* bf = (float*)(side == 0u ? buf_p : buf_m);
		uint8_t* bb = (uint8_t*)bf;
		const uint32_t layer = insert ? (side == 0u ? Nax-1u : 0u) : (side == 0u ? Nax-2u : 1u);
		for (uint32_t a = 0u; a < (uint32_t)A; a++) {
			uint32_t x, y, z; face_cell(cfg, axis, a, layer, &x, &y, &z);
			const uint64_t n = cell_index(cfg, x, y, z);
			if (!insert) { bf[a] = rho[n]; bf[A+a] = u[n]; bf[2u*A+a] = u[N+n]; bf[3u*A+a] = u[2u*N+n]; bb[16u*A+a] = flags[n]; }
			else { rho[n] = bf[a]; u[n] = bf[A+a]; u[N+n] = bf[2u*A+a]; u[2u*N+n] = bf[3u*A+a]; flags[n] = bb[16u*A+a]; }
		}
	}
}
/* kernel.cpp:2202-2223 */
void oracle_transfer_F(const oracle_cfg* cfg, int insert, uint32_t axis, void* buf_p, void* buf_m, float* F) {
	const uint64_t N = cells(cfg), A = face_area(cfg, axis);
	const uint32_t Nax = axis == 0u ? cfg->Nx : axis == 1u ? cfg->Ny : cfg->Nz;
	for (uint32_t side = 0u; side < 2u; side++) {
		float* bf = (float*)(side == 0u ? buf_p : buf_m);
		const uint32_t layer = insert ? (side == 0u ? Nax-1u : 0u) : (side == 0u ? Nax-2u : 1u);
		for (uint32_t a = 0u; a < (uint32_t)A; a++) {
			uint32_t x, y, z; face_cell(cfg, axis, a, layer, &x, &y, &z);
			const uint64_t n = cell_index(cfg, x, y, z);
			if (!insert) { bf[a] = F[n]; bf[A+a] = F[N+n]; bf[2u*A+a] = F[2u*N+n]; }
			else { F[n] = bf[a]; F[N+n] = bf[A+a]; F[2u*N+n] = bf[2u*A+a]; }
		}
	}
}


/* ------------------------------------------------------------------------------------------------
 * mesh voxelizer (kernel.cpp:2295-2390): one ray per cell of the domain face perpendicular to `direction`,
 * bidirectional Moeller-Trumbore against every triangle, distances of the hits ahead kept as truncated
 * 16-bit integers and sorted, then the lattice column is walked and cells between odd and even hits
 * are flagged.  params = the 16 floats the reference packs (lbm.cpp:351-374): triangle count (as bits),
 * bounding box x0 y0 z0 x1 y1 z1 (+-2 cells), rotation centre, linear velocity, rotational velocity.
 * Plain float arithmetic, no fused operations: dot = (ax*bx+ay*by)+az*bz, cross by components.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } vec3;
static vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static vec3 vsub(vec3 a, vec3 b) { return v3(a.x-b.x, a.y-b.y, a.z-b.z); }
static vec3 vadd(vec3 a, vec3 b) { return v3(a.x+b.x, a.y+b.y, a.z+b.z); }
static float vdot(vec3 a, vec3 b) { return a.x*b.x+a.y*b.y+a.z*b.z; }
static vec3 vcross(vec3 a, vec3 b) { return v3(a.y*b.z-a.z*b.y, a.z*b.x-a.x*b.z, a.x*b.y-a.y*b.x); }
static int clampi(int x, int lo, int hi) { return x < lo ? lo : x > hi ? hi : x; }
/* cell centre in global lattice coordinates: position() (kernel.cpp:834-836) plus the domain offset (kernel.cpp:2306) */
static vec3 cell_position(const oracle_cfg* cfg, const int* O, uint32_t x, uint32_t y, uint32_t z) {
	const vec3 p = v3((float)x+0.5f-0.5f*(float)cfg->Nx, (float)y+0.5f-0.5f*(float)cfg->Ny, (float)z+0.5f-0.5f*(float)cfg->Nz);
	const vec3 off = v3(0.5f*(float)((int)cfg->Nx+2*O[0])-0.5f, 0.5f*(float)((int)cfg->Ny+2*O[1])-0.5f, 0.5f*(float)((int)cfg->Nz+2*O[2])-0.5f);
	return vadd(p, off);
}
void oracle_voxelize_mesh(const oracle_cfg* cfg, const int* O, uint32_t direction, void* fi, float* u, uint8_t* flags, uint64_t t, uint8_t flag,
                          const float* p0, const float* p1, const float* p2, const float* params) {
	const uint64_t N = cells(cfg), A = face_area(cfg, direction);
	uint32_t triangle_number; memcpy(&triangle_number, &params[0], 4);
	const float x0 = params[1], y0 = params[2], z0 = params[3], x1 = params[4], y1 = params[5], z1 = params[6];
	const vec3 rc = v3(params[7], params[8], params[9]), lin = v3(params[10], params[11], params[12]), rot = v3(params[13], params[14], params[15]);
	const int Nd[3] = {(int)cfg->Nx, (int)cfg->Ny, (int)cfg->Nz};
	const float lo[3] = {x0, y0, z0}, hi[3] = {x1, y1, z1};
	for (uint32_t a = 0u; a < (uint32_t)A; a++) {
		uint32_t c[3];
		const uint32_t start = (uint32_t)clampi((int)lo[direction]-O[direction], 0, Nd[direction]-1);
		if (direction == 0u) { c[0] = start; c[1] = a%cfg->Ny; c[2] = a/cfg->Ny; }
		else if (direction == 1u) { c[0] = a/cfg->Nz; c[1] = start; c[2] = a%cfg->Nz; }
		else { c[0] = a%cfg->Nx; c[1] = a/cfg->Nx; c[2] = start; }
		const vec3 origin = cell_position(cfg, O, c[0], c[1], c[2]);
		const vec3 dirv = v3((float)(direction == 0u), (float)(direction == 1u), (float)(direction == 2u));
		const int outside = direction == 0u ? (origin.y < y0 || origin.z < z0 || origin.y >= y1 || origin.z >= z1)
		                  : direction == 1u ? (origin.x < x0 || origin.z < z0 || origin.x >= x1 || origin.z >= z1)
		                  : (origin.x < x0 || origin.y < y0 || origin.x >= x1 || origin.y >= y1);
		if (outside) continue;
		uint32_t ahead = 0u, behind = 0u;
		uint16_t dist[64];
		for (uint32_t i = 0u; i < 64u; i++) dist[i] = 0u;
		for (uint32_t i = 0u; i < triangle_number; i++) {
			const vec3 a0 = v3(p0[3u*i], p0[3u*i+1u], p0[3u*i+2u]), a1 = v3(p1[3u*i], p1[3u*i+1u], p1[3u*i+2u]), a2 = v3(p2[3u*i], p2[3u*i+1u], p2[3u*i+2u]);
			const vec3 e1 = vsub(a1, a0), e2 = vsub(a2, a0), w = vsub(origin, a0), h = vcross(dirv, e2), q = vcross(w, e1);
			const float g = vdot(e1, h), f = 1.0f/g, s = f*vdot(w, h), tt = f*vdot(dirv, q), d = f*vdot(e2, q);
			if (g != 0.0f && s >= 0.0f && s < 1.0f && tt >= 0.0f && s+tt < 1.0f) {
				if (d > 0.0f) { if (ahead < 64u && d < 65536.0f) dist[ahead] = (uint16_t)d; ahead++; }
				else behind++;
			}
		}
		const uint32_t kept = ahead < 64u ? ahead : 64u;
		for (uint32_t i = 1u; i < kept; i++) { /* ascending */
			const uint16_t v = dist[i]; uint32_t j = i;
			while (j > 0u && dist[j-1u] > v) { dist[j] = dist[j-1u]; j--; }
			dist[j] = v;
		}
		int inside = (ahead%2u) && (behind%2u);
		const int set_u = lin.x*lin.x+lin.y*lin.y+lin.z*lin.z+rot.x*rot.x+rot.y*rot.y+rot.z*rot.z > 0.0f;
		uint32_t next = (ahead%2u) != (behind%2u); /* parity disagreement: skip the first hit */
		const uint32_t h0 = c[direction];
		const uint32_t hmax = (uint32_t)clampi((int)hi[direction]-O[direction], 0, Nd[direction]);
		const uint32_t last = ahead-1u < 63u ? ahead-1u : 63u; /* ahead==0 wraps to 63 like the reference's min(intersections-1u, 63u) */
		const uint32_t hmesh = h0+(uint32_t)dist[last];
		for (uint32_t hcell = h0; hcell < hmax; hcell++) {
			while (next < ahead && hcell > h0+(uint32_t)dist[next < 63u ? next : 63u]) { inside = !inside; next++; }
			inside = inside && (next < ahead && hcell < hmesh);
			uint32_t cc[3] = {c[0], c[1], c[2]}; cc[direction] = hcell;
			const uint64_t n = cell_index(cfg, cc[0], cc[1], cc[2]);
			uint8_t fl = flags[n];
			const vec3 p = cell_position(cfg, O, cc[0], cc[1], cc[2]);
			const vec3 u_set = vadd(lin, vcross(vsub(rc, p), rot));
			if (inside) {
				fl = (uint8_t)((fl&~TYPE_BO)|flag);
				if (set_u) { u[n] = u_set.x; u[N+n] = u_set.y; u[2u*N+n] = u_set.z; }
			} else if ((fl&TYPE_BO) == TYPE_S) {
				const float ux = u[n], uy = u[N+n], uz = u[2u*N+n];
				if (ux == u_set.x && uy == u_set.y && uz == u_set.z) { /* the cell belonged to this mesh: release it */
					if (set_u) {
						uint64_t j[27]; neighbors(cfg, cc[0], cc[1], cc[2], j);
						float feq[27]; f_eq(cfg, 1.0f, ux, uy, uz, feq);
						store_f(cfg, fi, n, j, t, feq);
					}
					fl = (fl&TYPE_BO) == TYPE_MS ? (uint8_t)(fl&~TYPE_MS) : (uint8_t)(fl&~flag);
				}
			}
			flags[n] = fl;
		}
	}
}
/* kernel.cpp:2386-2390 */
void oracle_unvoxelize_mesh(const oracle_cfg* cfg, const int* O, uint8_t* flags, uint8_t flag, float x0, float y0, float z0, float x1, float y1, float z1) {
	const uint64_t N = cells(cfg);
	for (uint64_t n = 0u; n < N; n++) {
		uint32_t x, y, z; coordinates(cfg, n, &x, &y, &z);
		const vec3 p = cell_position(cfg, O, x, y, z);
		if (p.x >= x0-1.0f && p.y >= y0-1.0f && p.z >= z0-1.0f && p.x <= x1+1.0f && p.y <= y1+1.0f && p.z <= z1+1.0f) flags[n] &= (uint8_t)~flag;
	}
}

/* storage-format converters exposed so tests can pin them against numpy float16 and the reference */
/* ----------------------------------------------------------------------------------------------
 * Object sums: object_center_of_mass / object_force / object_torque (kernel.cpp:1929-1987) for ONE domain, as
 * the kernels define them: every cell of the domain's array whose flags equal `marker` (halo layers included),
 * positions measured from the centre of this domain's own box (kernel.cpp:834-836).  Work-groups of 64 cells
 * reduce by the pairwise tree of kernel.cpp:1938-1944; group sums are added in ascending group order (on a GPU
 * the atomics of kernel.cpp:1948-1951 land in no fixed order, so the last bits of the reference's own result
 * vary from run to run; ascending order is what oracle/_ref produces).  what: 0 positions, sum[3] = cell count
 * as integer bits (lbm.cpp:1522); 1 force; 2 torque about (cx, cy, cz).
 * ---------------------------------------------------------------------------------------------- */
void oracle_object_sum(const oracle_cfg* cfg, int what, const float* F, const uint8_t* flags, uint8_t marker, float cx, float cy, float cz, float* sum) {
	const uint64_t N = cells(cfg), groups = (N+63u)/64u;
	uint32_t total_cells = 0u;
	sum[0] = sum[1] = sum[2] = sum[3] = 0.0f;
	for (uint64_t g = 0u; g < groups; g++) {
		vec3 cache[64];
		uint32_t count[64];
		for (uint32_t lid = 0u; lid < 64u; lid++) {
			const uint64_t n = g*64u+lid;
			cache[lid] = v3(0.0f, 0.0f, 0.0f);
			count[lid] = 0u;
			if (n >= N || flags[n] != marker) continue;
			uint32_t x, y, z;
			coordinates(cfg, n, &x, &y, &z);
			const vec3 p = v3((float)x+0.5f-0.5f*(float)cfg->Nx, (float)y+0.5f-0.5f*(float)cfg->Ny, (float)z+0.5f-0.5f*(float)cfg->Nz);
			count[lid] = 1u;
			if (what == 0) cache[lid] = p;
			else {
				const vec3 f = v3(F[n], F[N+n], F[2u*N+n]);
				cache[lid] = what == 1 ? f : vcross(vsub(p, v3(cx, cy, cz)), f);
			}
		}
		for (uint32_t s = 1u; s < 64u; s *= 2u) {
			for (uint32_t lid = 0u; lid < 64u; lid += 2u*s) { cache[lid] = vadd(cache[lid], cache[lid+s]); count[lid] += count[lid+s]; }
		}
		if (what == 0) {
			if (count[0] > 0u) { sum[0] += cache[0].x; sum[1] += cache[0].y; sum[2] += cache[0].z; total_cells += count[0]; }
		} else {
			if (cache[0].x != 0.0f) sum[0] += cache[0].x;
			if (cache[0].y != 0.0f) sum[1] += cache[0].y;
			if (cache[0].z != 0.0f) sum[2] += cache[0].z;
		}
	}
	if (what == 0) memcpy(&sum[3], &total_cells, 4u);
}

float oracle_half_to_float(uint16_t h) { return half_to_float(h); }
uint16_t oracle_float_to_half(float x) { return float_to_half_rte(x); }
float oracle_fp16c_to_float(uint16_t h) { return fp16c_to_float(h); }
uint16_t oracle_float_to_fp16c(float x) { return float_to_fp16c(x); }
uint32_t oracle_transfers(uint32_t Q) { return transfers(Q); }
uint32_t oracle_dimensions(uint32_t Q) { return dimensions(Q); }
