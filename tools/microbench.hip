// Access-pattern ceilings on MI355X for the esoteric-pull DDF traffic (diagnostic tool, not part of the product).
// Measures, on a lattice of n^3 cells with Q SoA streams: (a) a plain float4 copy, (b) in-place read-modify-write
// of Q aligned streams with V cells per lane, (c) the same with the LBM's row/plane shifted neighbour slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
	for (size_t i = (size_t)blockIdx.x*blockDim.x+threadIdx.x; i < n4; i += (size_t)gridDim.x*blockDim.x) b[i] = a[i];
}
// copy with U independent 16-byte loads in flight per lane; NT uses non-temporal accesses
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT> __global__ __launch_bounds__(256) void k_copy_u(const float4* __restrict__ a_, float4* __restrict__ b_, size_t n4) {
	const f4* a = (const f4*)a_; f4* b = (f4*)b_;
	const size_t base = (size_t)blockIdx.x*blockDim.x*U+threadIdx.x;
	f4 t[U];
#pragma unroll
	for (int k = 0; k < U; k++) { const size_t i = base+(size_t)k*blockDim.x; if (i < n4) t[k] = NT ? __builtin_nontemporal_load(a+i) : a[i]; }
#pragma unroll
	for (int k = 0; k < U; k++) { const size_t i = base+(size_t)k*blockDim.x; if (i < n4) { if (NT) __builtin_nontemporal_store(t[k], b+i); else b[i] = t[k]; } }
}
template <int U> __global__ __launch_bounds__(256) void k_read_u(const float4* __restrict__ a, float* out, size_t n4) {
	const size_t base = (size_t)blockIdx.x*blockDim.x*U+threadIdx.x;
	float s = 0.0f;
#pragma unroll
	for (int k = 0; k < U; k++) { const size_t i = base+(size_t)k*blockDim.x; if (i < n4) { const float4 t = a[i]; s += t.x+t.y+t.z+t.w; } }
	if (s == 123.456f) out[0] = s;
}
template <int U> __global__ __launch_bounds__(256) void k_write_u(float4* __restrict__ b, size_t n4, float v) {
	const size_t base = (size_t)blockIdx.x*blockDim.x*U+threadIdx.x;
#pragma unroll
	for (int k = 0; k < U; k++) { const size_t i = base+(size_t)k*blockDim.x; if (i < n4) b[i] = make_float4(v, v, v, v); }
}
// in-place update of ONE stream: every 16-byte word is read and written back by the same lane
template <int U> __global__ __launch_bounds__(256) void k_inplace_u(float4* __restrict__ b, size_t n4, float v) {
	const size_t base = (size_t)blockIdx.x*blockDim.x*U+threadIdx.x;
	float4 t[U];
#pragma unroll
	for (int k = 0; k < U; k++) { const size_t i = base+(size_t)k*blockDim.x; if (i < n4) t[k] = b[i]; }
#pragma unroll
	for (int k = 0; k < U; k++) { const size_t i = base+(size_t)k*blockDim.x; if (i < n4) { t[k].x *= v; t[k].y += v; b[i] = t[k]; } }
}
template <int V> struct Vec { typedef float type __attribute__((ext_vector_type(V))); };
template <> struct Vec<1> { typedef float type; };

// MODE 0: all streams aligned at the own cell. MODE 1: odd streams of each pair at the neighbour (x+cx, y+cy, z+cz) like D3Q19.
template <int V, int MODE, bool NT, bool NTS = NT> __global__ __launch_bounds__(1024) void k_rmw(float* fi, float* fo, size_t N, unsigned Nx, unsigned Ny, unsigned Nz, unsigned xchunks, float s) {
	typedef typename Vec<V>::type vec;
	const unsigned chunk = blockIdx.x%xchunks, r = blockIdx.x/xchunks, z = r/Ny, y = r%Ny;
	const unsigned x = (chunk*blockDim.x+threadIdx.x)*V;
	if (x >= Nx) return;
	const unsigned yq[3] = {y, y+1 == Ny ? 0 : y+1, y == 0 ? Ny-1 : y-1}, zq[3] = {z, z+1 == Nz ? 0 : z+1, z == 0 ? Nz-1 : z-1};
	const size_t row0 = ((size_t)z*Ny+y)*Nx;
	constexpr int cx[9] = {1,0,0,1,1,0,1,1,0}, cy[9] = {0,1,0,1,0,1,-1,0,1}, cz[9] = {0,0,1,0,1,1,0,-1,-1};
	vec f[19];
	auto ld = [&](const float* p, unsigned xx, bool aligned) -> vec {
		vec t;
		if (aligned) { if (NT) t = __builtin_nontemporal_load((const vec*)(p+xx)); else t = *(const vec*)(p+xx); }
		else __builtin_memcpy(&t, p+xx, sizeof(vec));
		return t;
	};
	auto st = [&](float* p, unsigned xx, bool aligned, vec t) {
		if (aligned) { if (NTS) __builtin_nontemporal_store(t, (vec*)(p+xx)); else *(vec*)(p+xx) = t; }
		else __builtin_memcpy(p+xx, &t, sizeof(vec));
	};
	f[0] = ld(fi+row0, x, true);
#pragma unroll
	for (int k = 0; k < 9; k++) {
		f[1+2*k] = ld(fi+(size_t)(2+2*k)*N+row0, x, true);
		if (MODE == 0) f[2+2*k] = ld(fi+(size_t)(1+2*k)*N+row0, x, true);
		else {
			const float* p = fo+(size_t)(1+2*k)*N+((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx;
			if (cx[k] == 0) f[2+2*k] = ld(p, x, true);
			else if (x+V < Nx) f[2+2*k] = ld(p, x+1, false);
			else { vec t; for (int v = 0; v < V; v++) { unsigned xn = x+v+1; float e = p[xn == Nx ? 0 : xn]; if constexpr (V == 1) t = e; else t[v] = e; } f[2+2*k] = t; }
		}
	}
	vec sum = f[0];
#pragma unroll
	for (int i = 1; i < 19; i++) sum += f[i];
#pragma unroll
	for (int i = 0; i < 19; i++) f[i] = f[i]*s+sum*1e-3f;
	st(fo+row0, x, true, f[0]);
#pragma unroll
	for (int k = 0; k < 9; k++) {
		st(fo+(size_t)(2+2*k)*N+row0, x, true, f[2+2*k]);
		if (MODE == 0) st(fo+(size_t)(1+2*k)*N+row0, x, true, f[1+2*k]);
		else {
			float* p = fo+(size_t)(1+2*k)*N+((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx;
			if (cx[k] == 0) st(p, x, true, f[1+2*k]);
			else if (x+V < Nx) st(p, x+1, false, f[1+2*k]);
			else { for (int v = 0; v < V; v++) { unsigned xn = x+v+1; float e; if constexpr (V == 1) e = f[1+2*k]; else e = f[1+2*k][v]; p[xn == Nx ? 0 : xn] = e; } }
		}
	}
}


// 16-bit variant of k_rmw MODE 1: Q19 streams of ushort, V cells per lane; SHIFT 0: shifted slots element by element,
// SHIFT 1: shifted slots rebuilt from the lane's own aligned vector and the next lane's first element (DPP/shuffle),
// so that every access is an aligned V*2-byte vector.  WORK extra dependent FMAs per value emulate the collision cost.
template <int V, int SHIFT, int WORK> __global__ __launch_bounds__(256) void k_rmw16(unsigned short* fi, size_t N, unsigned Nx, unsigned Ny, unsigned Nz, unsigned xchunks, float s) {
	typedef unsigned short vec __attribute__((ext_vector_type(V)));
	const unsigned chunk = blockIdx.x%xchunks, r = blockIdx.x/xchunks, z = r/Ny, y = r%Ny;
	const unsigned x = (chunk*blockDim.x+threadIdx.x)*V;
	if (x >= Nx) return;
	const unsigned yq[3] = {y, y+1 == Ny ? 0 : y+1, y == 0 ? Ny-1 : y-1}, zq[3] = {z, z+1 == Nz ? 0 : z+1, z == 0 ? Nz-1 : z-1};
	const size_t row0 = ((size_t)z*Ny+y)*Nx;
	constexpr int cx[9] = {1,0,0,1,1,0,1,1,0}, cy[9] = {0,1,0,1,0,1,-1,0,1}, cz[9] = {0,0,1,0,1,1,0,-1,-1};
	vec f[19];
	f[0] = *(const vec*)(fi+row0+x);
#pragma unroll
	for (int k = 0; k < 9; k++) {
		f[1+2*k] = *(const vec*)(fi+(size_t)(2+2*k)*N+row0+x);
		const unsigned short* p = fi+(size_t)(1+2*k)*N+((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx;
		if (cx[k] == 0) f[2+2*k] = *(const vec*)(p+x);
		else if (SHIFT == 0) { vec t; for (int v = 0; v < V; v++) { const unsigned xn = x+v+1; t[v] = p[xn == Nx ? 0 : xn]; } f[2+2*k] = t; }
		else {
			const vec own = *(const vec*)(p+x);
			unsigned first = own[0];
			unsigned nxt = __shfl_down(first, 1);                       // first element of the next lane's vector
			if ((threadIdx.x&63u) == 63u || x+V >= Nx) { const unsigned xn = x+V; nxt = p[xn >= Nx ? 0 : xn]; }
			vec t; for (int v = 0; v+1 < V; v++) t[v] = own[v+1]; t[V-1] = (unsigned short)nxt;
			f[2+2*k] = t;
		}
	}
	float g[19][V];
	float sum[V];
#pragma unroll
	for (int v = 0; v < V; v++) { sum[v] = 0.0f; for (int i = 0; i < 19; i++) { g[i][v] = (float)f[i][v]; sum[v] += g[i][v]; } }
#pragma unroll
	for (int v = 0; v < V; v++) for (int i = 0; i < 19; i++) { float t = g[i][v]*s+sum[v]*1e-3f; for (int w = 0; w < WORK; w++) t = fmaf(t, s, 1e-3f); g[i][v] = t; }
	vec o[19];
#pragma unroll
	for (int i = 0; i < 19; i++) for (int v = 0; v < V; v++) o[i][v] = (unsigned short)g[i][v];
	*(vec*)(fi+row0+x) = o[0];
#pragma unroll
	for (int k = 0; k < 9; k++) {
		*(vec*)(fi+(size_t)(2+2*k)*N+row0+x) = o[2+2*k];
		unsigned short* p = fi+(size_t)(1+2*k)*N+((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx;
		const vec t = o[1+2*k];
		if (cx[k] == 0) *(vec*)(p+x) = t;
		else if (SHIFT == 0) { for (int v = 0; v < V; v++) { const unsigned xn = x+v+1; p[xn == Nx ? 0 : xn] = t[v]; } }
		else {
			// the aligned vector at x holds the cells x..x+V-1: element 0 comes from the previous lane's last value
			unsigned last = t[V-1];
			unsigned prev = __shfl_up(last, 1);
			const bool edge = (threadIdx.x&63u) == 0u;
			vec w; w[0] = (unsigned short)prev; for (int v = 1; v < V; v++) w[v] = t[v-1];
			if (!edge) *(vec*)(p+x) = w; else { for (int v = 1; v < V; v++) p[x+v] = w[v]; }
			if ((threadIdx.x&63u) == 63u || x+V >= Nx) { const unsigned xn = x+V; p[xn >= Nx ? 0 : xn] = t[V-1]; }
		}
	}
}

// Blocked layout: 64 consecutive cells x 19 DDFs form one contiguous 4864-byte block, addr(i, n) = ((n/64)*19+i)*64+n%64.
// Same slot pattern as k_rmw MODE 1 (odd slots of each pair at the neighbour cell), one cell per lane, in place.
__global__ __launch_bounds__(256) void k_rmw_blocked(float* fi, unsigned Nx, unsigned Ny, unsigned Nz, unsigned xchunks, float s) {
	const unsigned chunk = blockIdx.x%xchunks, r = blockIdx.x/xchunks, z = r/Ny, y = r%Ny;
	const unsigned x = chunk*blockDim.x+threadIdx.x;
	if (x >= Nx) return;
	const unsigned yq[3] = {y, y+1 == Ny ? 0 : y+1, y == 0 ? Ny-1 : y-1}, zq[3] = {z, z+1 == Nz ? 0 : z+1, z == 0 ? Nz-1 : z-1};
	constexpr int cx[9] = {1,0,0,1,1,0,1,1,0}, cy[9] = {0,1,0,1,0,1,-1,0,1}, cz[9] = {0,0,1,0,1,1,0,-1,-1};
	auto addr = [&](unsigned i, size_t n) -> size_t { return ((n>>6)*19u+i)*64u+(n&63u); };
	const size_t n = ((size_t)z*Ny+y)*Nx+x;
	const unsigned xp = x+1 == Nx ? 0 : x+1;
	float f[19];
	size_t a[19];
	a[0] = addr(0, n);
#pragma unroll
	for (int k = 0; k < 9; k++) {
		a[1+2*k] = addr(2+2*k, n);
		const size_t nb = ((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx+(cx[k] ? xp : x);
		a[2+2*k] = addr(1+2*k, nb);
	}
#pragma unroll
	for (int i = 0; i < 19; i++) f[i] = fi[a[i]];
	float sum = f[0];
#pragma unroll
	for (int i = 1; i < 19; i++) sum += f[i];
#pragma unroll
	for (int i = 0; i < 19; i++) f[i] = f[i]*s+sum*1e-3f;
	fi[a[0]] = f[0];
#pragma unroll
	for (int k = 0; k < 9; k++) { fi[a[1+2*k]] = f[2+2*k]; fi[a[2+2*k]] = f[1+2*k]; }
}

// k_rmw MODE 1, one cell per lane, with the stores (SCOPE_ST) and loads (SCOPE_LD) issued as relaxed atomics of a wider
// memory scope: 0 = ordinary access, 1 = agent scope, 2 = system scope (write-through / cache-bypassing coherence bits)
template <int SCOPE_LD, int SCOPE_ST> __global__ __launch_bounds__(256) void k_rmw_scope(float* fi, size_t N, unsigned Nx, unsigned Ny, unsigned Nz, unsigned xchunks, float s) {
	const unsigned chunk = blockIdx.x%xchunks, r = blockIdx.x/xchunks, z = r/Ny, y = r%Ny;
	const unsigned x = chunk*blockDim.x+threadIdx.x;
	if (x >= Nx) return;
	const unsigned yq[3] = {y, y+1 == Ny ? 0 : y+1, y == 0 ? Ny-1 : y-1}, zq[3] = {z, z+1 == Nz ? 0 : z+1, z == 0 ? Nz-1 : z-1};
	const size_t row0 = ((size_t)z*Ny+y)*Nx;
	constexpr int cx[9] = {1,0,0,1,1,0,1,1,0}, cy[9] = {0,1,0,1,0,1,-1,0,1}, cz[9] = {0,0,1,0,1,1,0,-1,-1};
	const unsigned xp = x+1 == Nx ? 0 : x+1;
	auto ld = [&](float* p) -> float {
		if constexpr (SCOPE_LD == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else if constexpr (SCOPE_LD == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		else return *p;
	};
	auto st = [&](float* p, float v) {
		if constexpr (SCOPE_ST == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else if constexpr (SCOPE_ST == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		else *p = v;
	};
	float* a[19];
	a[0] = fi+row0+x;
#pragma unroll
	for (int k = 0; k < 9; k++) {
		a[1+2*k] = fi+(size_t)(2+2*k)*N+row0+x;
		a[2+2*k] = fi+(size_t)(1+2*k)*N+((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx+(cx[k] ? xp : x);
	}
	float f[19];
#pragma unroll
	for (int i = 0; i < 19; i++) f[i] = ld(a[i]);
	float sum = f[0];
#pragma unroll
	for (int i = 1; i < 19; i++) sum += f[i];
#pragma unroll
	for (int i = 0; i < 19; i++) f[i] = f[i]*s+sum*1e-3f;
	st(a[0], f[0]);
#pragma unroll
	for (int k = 0; k < 9; k++) { st(a[1+2*k], f[2+2*k]); st(a[2+2*k], f[1+2*k]); }
}

template <class F> float time_ms(F launch, int reps) {
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	launch(); launch();
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; i++) launch();
	CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
	float ms; CHECK(hipEventElapsedTime(&ms, a, b));
	return ms/reps;
}
template <int V, int MODE, bool NT> void run(float* fi, size_t N, unsigned n, unsigned block, size_t pad, unsigned blocks_per_cu = 0) {
	float* fo = fi; (void)fo;
	const unsigned lanes = n/V, b = block > lanes ? lanes : block, xchunks = (lanes+b-1)/b;
	const unsigned lds = blocks_per_cu ? (160u*1024u/blocks_per_cu)&~255u : 0u; // dynamic LDS only throttles residency
	const float ms = time_ms([&]() { hipLaunchKernelGGL((k_rmw<V, MODE, NT>), dim3(xchunks*n*n), dim3(b), lds, 0, fi, fi, N+pad, n, n, n, xchunks, 0.999f); }, 10);
	printf("rmw19 V=%d mode=%d nt=%d block=%3u pad=%zu blocks/CU<=%u: %.3f ms  %.0f GB/s (%.1f%%)  %.0f MLUPs-equivalent\n", V, MODE, (int)NT, b, pad, blocks_per_cu, ms, 152.0*N/ms/1e6, 152.0*N/ms/1e6/80, N/ms/1e3);
	fflush(stdout);
}
int main(int argc, char** argv) {
	const unsigned n = argc > 1 ? atoi(argv[1]) : 512;
	const size_t N = (size_t)n*n*n, padmax = 1u<<20;
	const bool oop = argc > 2 && atoi(argv[2]) != 0; // second argument 1: streams are written to a second buffer (two-lattice pattern)
	float* fi;
	{ // MB_ALLOC=uncached | finegrained | contiguous selects the memory type of the lattice (default: ordinary hipMalloc)
		const char* kind = getenv("MB_ALLOC");
		const size_t bytes = (N+padmax)*19*sizeof(float)*(oop ? 2 : 1);
		if (kind && !strcmp(kind, "uncached")) CHECK(hipExtMallocWithFlags((void**)&fi, bytes, hipDeviceMallocUncached));
		else if (kind && !strcmp(kind, "finegrained")) CHECK(hipExtMallocWithFlags((void**)&fi, bytes, hipDeviceMallocFinegrained));
		else if (kind && !strcmp(kind, "contiguous")) CHECK(hipExtMallocWithFlags((void**)&fi, bytes, hipDeviceMallocContiguous));
		else CHECK(hipMalloc(&fi, bytes));
		printf("allocation: %s\n", kind ? kind : "default");
	}
	float* fo = oop ? fi+(N+padmax)*19 : fi;
	CHECK(hipMemset(fi, 0, (N+padmax)*19*sizeof(float)));
	{
		const size_t n4 = N*19/4/2;
		const float ms = time_ms([&]() { hipLaunchKernelGGL(k_copy, dim3(256*8), dim3(256), 0, 0, (const float4*)fi, (float4*)fi+n4, n4); }, 10);
		printf("float4 copy: %.3f ms %.0f GB/s (%.1f%% of 8 TB/s)\n", ms, 2.0*n4*16/ms/1e6, 2.0*n4*16/ms/1e6/80);
	}
	{
		const size_t n4 = N*19/4/2;
		float4* a = (float4*)fi; float4* b = (float4*)fi+n4;
		auto rep = [&](const char* name, float ms, double bytes) { printf("%-28s %.3f ms %.0f GB/s (%.1f%% of 8 TB/s)\n", name, ms, bytes/ms/1e6, bytes/ms/1e6/80); fflush(stdout); };
		rep("copy U=4", time_ms([&]() { hipLaunchKernelGGL((k_copy_u<4, false>), dim3((n4+1023)/1024), dim3(256), 0, 0, a, b, n4); }, 10), 2.0*n4*16);
		rep("copy U=8", time_ms([&]() { hipLaunchKernelGGL((k_copy_u<8, false>), dim3((n4+2047)/2048), dim3(256), 0, 0, a, b, n4); }, 10), 2.0*n4*16);
		rep("copy U=8 nt", time_ms([&]() { hipLaunchKernelGGL((k_copy_u<8, true>), dim3((n4+2047)/2048), dim3(256), 0, 0, a, b, n4); }, 10), 2.0*n4*16);
		rep("copy U=1", time_ms([&]() { hipLaunchKernelGGL((k_copy_u<1, false>), dim3((n4+255)/256), dim3(256), 0, 0, a, b, n4); }, 10), 2.0*n4*16);
		rep("read U=8", time_ms([&]() { hipLaunchKernelGGL((k_read_u<8>), dim3((2*n4+2047)/2048), dim3(256), 0, 0, a, (float*)b, 2*n4); }, 10), 2.0*n4*16);
		rep("write U=8", time_ms([&]() { hipLaunchKernelGGL((k_write_u<8>), dim3((2*n4+2047)/2048), dim3(256), 0, 0, a, 2*n4, 0.0f); }, 10), 2.0*n4*16);
		rep("inplace U=8", time_ms([&]() { hipLaunchKernelGGL((k_inplace_u<8>), dim3((n4+2047)/2048), dim3(256), 0, 0, a, n4, 1.0f); }, 10), 2.0*n4*16);
		rep("inplace U=2", time_ms([&]() { hipLaunchKernelGGL((k_inplace_u<2>), dim3((n4+511)/512), dim3(256), 0, 0, a, n4, 1.0f); }, 10), 2.0*n4*16);
		CHECK(hipMemset(fi, 0, (N+padmax)*19*sizeof(float)));
	}
	CHECK(hipMemset(fi, 0, (N+padmax)*19*sizeof(float)));
	if (argc > 3 && !strcmp(argv[3], "scope")) { // memory scope of the accesses: ordinary, agent, system
		auto one = [&](auto kern, const char* what) {
			for (unsigned k : {2u, 3u}) {
				const unsigned b = 256u, xchunks = (n+b-1)/b, lds = (160u*1024u/k)&~255u;
				const float ms = time_ms([&]() { hipLaunchKernelGGL(kern, dim3(xchunks*n*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, 0.999f); }, 10);
				printf("scope %-28s blocks/CU<=%u: %.3f ms  %.0f GB/s (%.1f%%)\n", what, k, ms, 152.0*N/ms/1e6, 152.0*N/ms/1e6/80); fflush(stdout);
			}
		};
		one(k_rmw_scope<0, 0>, "load plain, store plain"); one(k_rmw_scope<0, 1>, "load plain, store agent"); one(k_rmw_scope<0, 2>, "load plain, store system");
		one(k_rmw_scope<1, 0>, "load agent, store plain"); one(k_rmw_scope<2, 0>, "load system, store plain"); one(k_rmw_scope<1, 1>, "load agent, store agent"); one(k_rmw_scope<2, 2>, "load system, store system");
		return 0;
	}
	if (argc > 3 && !strcmp(argv[3], "blocked")) { // blocked (64 cells x 19 DDFs) layout against the stream layout, one cell per lane
		for (unsigned k : {0u, 2u, 3u, 4u, 6u, 8u}) {
			const unsigned b = 256u, xchunks = (n+b-1)/b;
			const unsigned lds = k ? (160u*1024u/k)&~255u : 0u;
			const float ms = time_ms([&]() { hipLaunchKernelGGL(k_rmw_blocked, dim3(xchunks*n*n), dim3(b), lds, 0, fi, n, n, n, xchunks, 0.999f); }, 10);
			printf("blocked 64x19 layout blocks/CU<=%u: %.3f ms  %.0f GB/s (%.1f%%)\n", k, ms, 152.0*N/ms/1e6, 152.0*N/ms/1e6/80); fflush(stdout);
			run<1, 1, false>(fi, N, n, 256u, 0, k);
		}
		return 0;
	}
	if (argc > 3 && !strcmp(argv[3], "pad")) { // stream-stride padding sweep for the FP32 one-cell-per-lane pattern
		for (size_t pad : {(size_t)0, (size_t)32, (size_t)64, (size_t)128, (size_t)256, (size_t)512, (size_t)1024, (size_t)2048, (size_t)4096, (size_t)4160, (size_t)8192, (size_t)8256, (size_t)16384, (size_t)16448, (size_t)65536, (size_t)65600, (size_t)262144, (size_t)262208, (size_t)524288, (size_t)1048576-64})
			for (unsigned k : {2u, 3u}) run<1, 1, false>(fi, N, n, 256u, pad, k);
		return 0;
	}
	if (argc > 3) { // third argument: 16-bit stream sweep only
		auto sweep16 = [&](auto kern, int V, int shift, int work) {
			for (unsigned k : {0u, 2u, 3u, 4u, 6u}) {
				const unsigned lanes = n/V, b = 256u > lanes ? lanes : 256u, xchunks = (lanes+b-1)/b;
				const unsigned lds = k ? (160u*1024u/k)&~255u : 0u;
				const float ms = time_ms([&]() { hipLaunchKernelGGL(kern, dim3(xchunks*n*n), dim3(b), lds, 0, (unsigned short*)fi, N, n, n, n, xchunks, 0.999f); }, 6);
				printf("rmw16 V=%d shift=%d work=%d blocks/CU<=%u: %.3f ms %5.0f GB/s (%.1f%%)\n", V, shift, work, k, ms, 76.0*N/ms/1e6, 76.0*N/ms/1e6/80); fflush(stdout);
			}
		};
		sweep16(k_rmw16<1, 0, 0>, 1, 0, 0); sweep16(k_rmw16<2, 0, 0>, 2, 0, 0); sweep16(k_rmw16<2, 1, 0>, 2, 1, 0); sweep16(k_rmw16<4, 0, 0>, 4, 0, 0); sweep16(k_rmw16<4, 1, 0>, 4, 1, 0);
		sweep16(k_rmw16<2, 0, 8>, 2, 0, 8); sweep16(k_rmw16<2, 1, 8>, 2, 1, 8); sweep16(k_rmw16<2, 0, 16>, 2, 0, 16); sweep16(k_rmw16<2, 1, 16>, 2, 1, 16);
		return 0;
	}
	struct R { float gbs; char name[96]; };
	std::vector<R> res;
	auto sweep = [&](auto kern, int V, int ntl, int nts) {
		for (unsigned block : {128u, 256u}) for (unsigned k : {0u, 1u, 2u, 3u, 4u, 6u}) for (size_t pad : {(size_t)0, (size_t)4160}) {
			const unsigned lanes = n/V, b = block > lanes ? lanes : block, xchunks = (lanes+b-1)/b;
			const unsigned lds = k ? (160u*1024u/k)&~255u : 0u;
			const float ms = time_ms([&]() { hipLaunchKernelGGL(kern, dim3(xchunks*n*n), dim3(b), lds, 0, fi, fo, N+pad, n, n, n, xchunks, 0.999f); }, 6);
			R r; r.gbs = 152.0*N/ms/1e6;
			snprintf(r.name, sizeof(r.name), "%sV=%d ntload=%d ntstore=%d block=%u blocks/CU<=%u pad=%zu", oop ? "two-buffer " : "", V, ntl, nts, b, k, pad);
			res.push_back(r);
		}
	};
	sweep(k_rmw<1, 1, false, false>, 1, 0, 0); sweep(k_rmw<1, 1, true, false>, 1, 1, 0); sweep(k_rmw<1, 1, false, true>, 1, 0, 1); sweep(k_rmw<1, 1, true, true>, 1, 1, 1);
	sweep(k_rmw<2, 1, false, false>, 2, 0, 0); sweep(k_rmw<2, 1, true, false>, 2, 1, 0); sweep(k_rmw<2, 1, false, true>, 2, 0, 1); sweep(k_rmw<2, 1, true, true>, 2, 1, 1);
	sweep(k_rmw<4, 1, false, false>, 4, 0, 0); sweep(k_rmw<4, 1, true, false>, 4, 1, 0); sweep(k_rmw<4, 1, false, true>, 4, 0, 1); sweep(k_rmw<4, 1, true, true>, 4, 1, 1);
	std::sort(res.begin(), res.end(), [](const R& a, const R& b) { return a.gbs > b.gbs; });
	for (size_t i = 0; i < res.size(); i++) printf("%5.0f GB/s (%.1f%%)  %s\n", res[i].gbs, res[i].gbs/80, res[i].name);
	return 0;
}
