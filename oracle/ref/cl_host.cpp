// Test infrastructure (oracle/_ref build, step 4): host-side environment for the reference's OpenCL C
// kernels after clang -x cl compiled them for x86-64.  Two parts, both ours:
//  (1) the OpenCL C language built-ins those kernels call, defined as the OpenCL C specification
//      describes them (single-precision, round-to-nearest);
//  (2) a driver that runs a 1-D NDRange by looping the global id, optionally split over std::threads
//      (cells are independent within one kernel launch, FluidX3D-master/src/kernel.cpp:1481-1485);
//      kernels that call barrier() (the object_* reductions, kernel.cpp:1929-1987) run one work-group at a
//      time on 64 threads that meet at a pthread barrier, groups in ascending order.
// The kernels' signatures are those of FluidX3D-master/src/kernel.cpp:1364, 1470, 1812, 1897, 1911,
// 2148, 2154, 2175, 2181, 2212, 2218.  Grid size, velocity set and feature set are baked into the
// kernel object; the REF_* macros given on the command line mirror them for this file.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <pthread.h>
#include <thread>
#include <vector>

typedef float float3 __attribute__((ext_vector_type(3)));
typedef unsigned int uint;
typedef unsigned long ulong;
typedef unsigned char uchar;

static thread_local ulong tl_global_id = 0ul;

// ---------------------------------------------------------------- (1) OpenCL C built-ins
#define BUILTIN(mangled) __asm__(mangled) __attribute__((visibility("hidden")))
ulong cl_get_global_id(uint) BUILTIN("_Z13get_global_idj");
ulong cl_get_local_id(uint) BUILTIN("_Z12get_local_idj");
void cl_barrier(uint) BUILTIN("_Z7barrierj");
float cl_fma(float, float, float) BUILTIN("_Z3fmafff");
uint cl_min(uint, uint) BUILTIN("_Z3minjj");
float cl_acos(float) BUILTIN("_Z4acosf");
float cl_fabs(float) BUILTIN("_Z4fabsf");
float cl_fmod(float, float) BUILTIN("_Z4fmodff");
float cl_sign(float) BUILTIN("_Z4signf");
float cl_sqrt(float) BUILTIN("_Z4sqrtf");
float cl_clampf(float, float, float) BUILTIN("_Z5clampfff");
int cl_clampi(int, int, int) BUILTIN("_Z5clampiii");
float cl_dot(float3, float3) BUILTIN("_Z3dotDv3_fS_");
float3 cl_cross(float3, float3) BUILTIN("_Z5crossDv3_fS_");
float cl_length(float3) BUILTIN("_Z6lengthDv3_f");
float cl_vload_half(ulong, const _Float16*) BUILTIN("_Z10vload_halfmPU9CLgenericKDh");
void cl_vstore_half_rte(float, ulong, _Float16*) BUILTIN("_Z15vstore_half_rtefmPU9CLgenericDh");
uint cl_atomic_add(volatile uint*, uint) BUILTIN("_Z10atomic_addPU8CLglobalVjj");
float cl_atomic_xchg(volatile float*, float) BUILTIN("_Z11atomic_xchgPU8CLglobalVff");

ulong cl_get_global_id(uint) { return tl_global_id; }
ulong cl_get_local_id(uint) { return tl_global_id % 64ul; }
static pthread_barrier_t group_barrier; // the 64 work-items of the one work-group in flight (ndrange_groups)
void cl_barrier(uint) { pthread_barrier_wait(&group_barrier); }
float cl_fma(float a, float b, float c) { return std::fmaf(a, b, c); }
uint cl_min(uint a, uint b) { return a < b ? a : b; }
float cl_acos(float x) { return std::acos(x); }
float cl_fabs(float x) { return std::fabs(x); }
float cl_fmod(float x, float y) { return std::fmod(x, y); }
float cl_sign(float x) { return x > 0.0f ? 1.0f : x < 0.0f ? -1.0f : 0.0f; }
float cl_sqrt(float x) { return std::sqrt(x); }
float cl_clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
int cl_clampi(int x, int lo, int hi) { return x < lo ? lo : x > hi ? hi : x; }
float cl_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float3 cl_cross(float3 a, float3 b) { return (float3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float cl_length(float3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
float cl_vload_half(ulong offset, const _Float16* p) { return (float)p[offset]; }                 // IEEE half -> float, exact
void cl_vstore_half_rte(float x, ulong offset, _Float16* p) { p[offset] = (_Float16)x; }        // round to nearest even
uint cl_atomic_add(volatile uint* p, uint v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
float cl_atomic_xchg(volatile float* p, float v) {
	uint nv, ov;
	std::memcpy(&nv, &v, 4);
	ov = __atomic_exchange_n((volatile uint*)p, nv, __ATOMIC_RELAXED);
	float r;
	std::memcpy(&r, &ov, 4);
	return r;
}

// ---------------------------------------------------------------- the reference's kernels (C symbols)
#if REF_FPXX_BYTES == 4
typedef float fpxx;
#else
typedef unsigned short fpxx;
#endif
#ifdef REF_FORCE_FIELD
#define F_PARAM , const float* F
#define F_ARG , F
#else
#define F_PARAM
#define F_ARG
#endif
extern "C" {
void initialize(fpxx* fi, const float* rho, float* u, uchar* flags);
void stream_collide(fpxx* fi, float* rho, float* u, uchar* flags, ulong t, float fx, float fy, float fz F_PARAM);
void update_fields(const fpxx* fi, float* rho, float* u, const uchar* flags, ulong t, float fx, float fy, float fz F_PARAM);
#ifdef REF_FORCE_FIELD
void update_force_field(const fpxx* fi, const uchar* flags, ulong t, float* F);
void reset_force_field(float* F);
void transfer_extract_F(uint direction, ulong t, float* p, float* m, const float* F);
void transfer__insert_F(uint direction, ulong t, const float* p, const float* m, float* F);
#endif
#ifdef REF_MOVING_BOUNDARIES
void update_moving_boundaries(const float* u, uchar* flags);
#endif
#ifdef REF_FORCE_FIELD
void object_center_of_mass(const uchar* flags, uchar flag_marker, volatile float* object_sum);
void object_force(const float* F, const uchar* flags, uchar flag_marker, volatile float* object_sum);
void object_torque(const float* F, const uchar* flags, uchar flag_marker, float cx, float cy, float cz, volatile float* object_sum);
#endif
void voxelize_mesh(uint direction, fpxx* fi, float* u, uchar* flags, ulong t, uchar flag, const float* p0, const float* p1, const float* p2, const float* bbu);
void unvoxelize_mesh(uchar* flags, uchar flag, float x0, float y0, float z0, float x1, float y1, float z1);
void transfer_extract_fi(uint direction, ulong t, fpxx* p, fpxx* m, const fpxx* fi);
void transfer__insert_fi(uint direction, ulong t, const fpxx* p, const fpxx* m, fpxx* fi);
void transfer_extract_rho_u_flags(uint direction, ulong t, char* p, char* m, const float* rho, const float* u, const uchar* flags);
void transfer__insert_rho_u_flags(uint direction, ulong t, const char* p, const char* m, float* rho, float* u, uchar* flags);
}

// ---------------------------------------------------------------- (2) NDRange driver
template <typename K> static void ndrange(const ulong range, const int threads, K kernel) {
	const ulong global = (range + 63ul) / 64ul * 64ul; // global size rounded up to the work-group size (FluidX3D-master/src/opencl.hpp:655-659)
	if (threads <= 1) {
		for (ulong n = 0ul; n < global; n++) { tl_global_id = n; kernel(); }
		return;
	}
	std::vector<std::thread> pool;
	for (int k = 0; k < threads; k++) {
		const ulong a = global * (ulong)k / (ulong)threads, b = global * (ulong)(k + 1) / (ulong)threads;
		pool.emplace_back([=]() { for (ulong n = a; n < b; n++) { tl_global_id = n; kernel(); } });
	}
	for (auto& th : pool) th.join();
}

// Work-groups of 64 one after the other (the kernels' local arrays are one static array each on this target), the 64
// work-items of a group on 64 threads so that barrier() has OpenCL's meaning.  Group sums therefore reach the kernels'
// atomics in ascending group order, which makes the float result reproducible.
template <typename K> static void ndrange_groups(const ulong range, K kernel) {
	const ulong groups = (range + 63ul) / 64ul;
	pthread_barrier_init(&group_barrier, nullptr, 64u);
	std::vector<std::thread> pool;
	for (ulong lid = 0ul; lid < 64ul; lid++) {
		pool.emplace_back([=]() {
			for (ulong g = 0ul; g < groups; g++) {
				tl_global_id = g * 64ul + lid;
				kernel();
				pthread_barrier_wait(&group_barrier); // the group's atomics are done before the next group reuses local memory
			}
		});
	}
	for (auto& th : pool) th.join();
	pthread_barrier_destroy(&group_barrier);
}

static const ulong N = (ulong)REF_NX * (ulong)REF_NY * (ulong)REF_NZ;
static ulong area(const uint direction) {
	const ulong A[3] = {(ulong)REF_NY * REF_NZ, (ulong)REF_NZ * REF_NX, (ulong)REF_NX * REF_NY};
	return A[direction];
}
static uint feature_mask() {
	uint m = 0u;
#ifdef REF_UPDATE_FIELDS
	m |= 1u;
#endif
#ifdef REF_VOLUME_FORCE
	m |= 2u;
#endif
#ifdef REF_MOVING_BOUNDARIES
	m |= 4u;
#endif
#ifdef REF_EQUILIBRIUM_BOUNDARIES
	m |= 8u;
#endif
#ifdef REF_FORCE_FIELD
	m |= 16u;
#endif
#ifdef REF_SUBGRID
	m |= 32u;
#endif
	return m;
}

#define API extern "C" __attribute__((visibility("default")))
API void ref_info(int* out) { // Nx Ny Nz Dx Dy Dz Q transfers fpxx_bytes features
	const int v[10] = {REF_NX, REF_NY, REF_NZ, REF_DX, REF_DY, REF_DZ, REF_Q, REF_TRANSFERS, REF_FPXX_BYTES, (int)feature_mask()};
	for (int i = 0; i < 10; i++) out[i] = v[i];
}
API void ref_initialize(void* fi, const float* rho, float* u, uchar* flags, int threads) {
	ndrange(N, threads, [=]() { initialize((fpxx*)fi, rho, u, flags); });
}
API void ref_stream_collide(void* fi, float* rho, float* u, uchar* flags, ulong t, float fx, float fy, float fz, const float* F, int threads) {
	(void)F;
	ndrange(N, threads, [=]() { stream_collide((fpxx*)fi, rho, u, flags, t, fx, fy, fz F_ARG); });
}
API void ref_update_fields(const void* fi, float* rho, float* u, const uchar* flags, ulong t, float fx, float fy, float fz, const float* F, int threads) {
	(void)F;
	ndrange(N, threads, [=]() { update_fields((const fpxx*)fi, rho, u, flags, t, fx, fy, fz F_ARG); });
}
API int ref_update_force_field(const void* fi, const uchar* flags, ulong t, float* F, int threads) {
#ifdef REF_FORCE_FIELD
	ndrange(N, threads, [=]() { update_force_field((const fpxx*)fi, flags, t, F); });
	return 0;
#else
	(void)fi; (void)flags; (void)t; (void)F; (void)threads;
	return -1;
#endif
}
API int ref_reset_force_field(float* F, int threads) {
#ifdef REF_FORCE_FIELD
	ndrange(N, threads, [=]() { reset_force_field(F); });
	return 0;
#else
	(void)F; (void)threads;
	return -1;
#endif
}
API int ref_update_moving_boundaries(const float* u, uchar* flags, int threads) {
#ifdef REF_MOVING_BOUNDARIES
	ndrange(N, threads, [=]() { update_moving_boundaries(u, flags); });
	return 0;
#else
	(void)u; (void)flags; (void)threads;
	return -1;
#endif
}
// what: 0 object_center_of_mass (sum[3] holds the cell count as uint bits), 1 object_force, 2 object_torque
API int ref_object_sum(int what, const float* F, const uchar* flags, uchar flag_marker, float cx, float cy, float cz, float* sum) {
#ifdef REF_FORCE_FIELD
	sum[0] = sum[1] = sum[2] = sum[3] = 0.0f; // LBM_Domain::enqueue_object_* reset (FluidX3D-master/src/lbm.cpp:269-271)
	if (what == 0) ndrange_groups(N, [=]() { object_center_of_mass(flags, flag_marker, sum); });
	else if (what == 1) ndrange_groups(N, [=]() { object_force(F, flags, flag_marker, sum); });
	else ndrange_groups(N, [=]() { object_torque(F, flags, flag_marker, cx, cy, cz, sum); });
	return 0;
#else
	(void)what; (void)F; (void)flags; (void)flag_marker; (void)cx; (void)cy; (void)cz; (void)sum;
	return -1;
#endif
}
API void ref_voxelize_mesh(uint direction, void* fi, float* u, uchar* flags, ulong t, uchar flag, const float* p0, const float* p1, const float* p2, const float* bbu) {
	ndrange(area(direction), 1, [=]() { voxelize_mesh(direction, (fpxx*)fi, u, flags, t, flag, p0, p1, p2, bbu); });
}
API void ref_unvoxelize_mesh(uchar* flags, uchar flag, float x0, float y0, float z0, float x1, float y1, float z1) {
	ndrange(N, 1, [=]() { unvoxelize_mesh(flags, flag, x0, y0, z0, x1, y1, z1); });
}
API void ref_transfer_fi(int insert, uint direction, ulong t, void* p, void* m, void* fi) {
	if (insert) ndrange(area(direction), 1, [=]() { transfer__insert_fi(direction, t, (const fpxx*)p, (const fpxx*)m, (fpxx*)fi); });
	else ndrange(area(direction), 1, [=]() { transfer_extract_fi(direction, t, (fpxx*)p, (fpxx*)m, (const fpxx*)fi); });
}
API void ref_transfer_rho_u_flags(int insert, uint direction, ulong t, void* p, void* m, float* rho, float* u, uchar* flags) {
	if (insert) ndrange(area(direction), 1, [=]() { transfer__insert_rho_u_flags(direction, t, (const char*)p, (const char*)m, rho, u, flags); });
	else ndrange(area(direction), 1, [=]() { transfer_extract_rho_u_flags(direction, t, (char*)p, (char*)m, rho, u, flags); });
}
API int ref_transfer_F(int insert, uint direction, ulong t, void* p, void* m, float* F) {
#ifdef REF_FORCE_FIELD
	if (insert) ndrange(area(direction), 1, [=]() { transfer__insert_F(direction, t, (const float*)p, (const float*)m, F); });
	else ndrange(area(direction), 1, [=]() { transfer_extract_F(direction, t, (float*)p, (float*)m, F); });
	return 0;
#else
	(void)insert; (void)direction; (void)t; (void)p; (void)m; (void)F;
	return -1;
#endif
}
