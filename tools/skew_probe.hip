// Diagnostic (not part of the product): what do two time steps in ONE launch gain, when the second step trails the
// first by a few lattice planes so that it finds the first step's DDFs in the 256 MiB Infinity Cache?
// The kernel is the in-place 19-stream access pattern of the FP32 step kernel (one cell per lane, one slot of each pair
// at the +c neighbour, the roles of the two streams of a pair alternating from pass to pass like esoteric-pull) with a
// trivial update that mixes the 19 values, so that pass 2 of a cell depends on pass 1 of its neighbours.
// The grid is a list of plane tasks: pass 1 runs `lag`+2 planes ahead, then pass 2 plane p alternates with pass 1 plane
// p+lag, and the planes next to the periodic seam (pass 2 of planes 0 and 1) come last.
// SYNC 0: no dependencies enforced (bandwidth only, values meaningless).
// SYNC 1: per-plane completion counters, release/acquire fences at agent scope (L2 write-back and invalidate): correct, 12 % of 8 TB/s.
// SYNC 2: per-plane completion counters; pass 1 stores write-through (agent scope), pass 2 loads at agent scope (past the L1).
// SYNC 3: per-plane completion counters; pass 1 stores write-through, pass 2 invalidates its L1 after the wait and loads plainly.
// SYNC 1 to 3 are checked bit for bit against two plain launches.
// usage: skew_probe [n] [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr unsigned SUB = 8; // completion counters per plane (spreads the atomics over addresses), 64 bytes apart

template <int SYNC> __global__ __launch_bounds__(256) void k_skew(float* fi, size_t N, unsigned Nx, unsigned Ny, unsigned Nz, unsigned xchunks, unsigned lag, unsigned passes, unsigned odd0,
		float s, unsigned* done, unsigned* error) {
	const unsigned per_plane = xchunks*Ny;
	const unsigned task = blockIdx.x/per_plane, in_plane = blockIdx.x%per_plane;
	unsigned z, pass = 0u;
	if (passes == 1u) z = task;
	else if (task < lag+2u) z = task;                                      // pass 1 runs ahead
	else if (task < lag+2u+2u*(Nz-lag-2u)) { const unsigned k = task-lag-2u; pass = (k&1u)^1u; z = pass ? 2u+k/2u : lag+2u+k/2u; } // pass 2 plane 2+j, then pass 1 plane lag+2+j
	else { pass = 1u; z = Nz-lag+(task-(lag+2u+2u*(Nz-lag-2u))); if (z >= Nz) z -= Nz; } // pass 2: planes Nz-lag .. Nz-1, 0, 1
	const unsigned odd = odd0^pass;
	if (SYNC != 0 && pass == 1u) { // wait for pass 1 on the planes z-2 .. z+2
		if (threadIdx.x < 64u) {
			const unsigned lane = threadIdx.x;
			const unsigned zz = (z+Nz-2u+(lane < 5u ? lane : 0u))%Nz;
			unsigned spins = 0u;
			while (true) {
				bool ok = true;
				if (lane < 5u) ok = __hip_atomic_load(done+zz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= SUB; // all shards of the plane have reported
				if (__ballot(ok) == ~0ull) break;
				__builtin_amdgcn_s_sleep(8);
				if (++spins > 4000000u) { if (lane == 0u) *error = 1u; break; } // never reached when workgroups are dispatched in order
			}
			if (SYNC == 1 || SYNC == 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // one invalidate of the CU's L1
		}
		__syncthreads();
	}
	const unsigned chunk = in_plane%xchunks, y = in_plane/xchunks;
	const unsigned x = chunk*blockDim.x+threadIdx.x;
	if (x < Nx) {
		const unsigned yq[3] = {y, y+1 == Ny ? 0 : y+1, y == 0 ? Ny-1 : y-1}, zq[3] = {z, z+1 == Nz ? 0 : z+1, z == 0 ? Nz-1 : z-1};
		const size_t row0 = ((size_t)z*Ny+y)*Nx;
		constexpr int cx[9] = {1,0,0,1,1,0,1,1,0}, cy[9] = {0,1,0,1,0,1,-1,0,1}, cz[9] = {0,0,1,0,1,1,0,-1,-1};
		const unsigned xp = x+1 == Nx ? 0 : x+1;
		float* a[19];
		a[0] = fi+row0+x;
#pragma unroll
		for (int k = 0; k < 9; k++) {
			const unsigned sA = odd ? 1+2*k : 2+2*k, sB = odd ? 2+2*k : 1+2*k;
			a[1+2*k] = fi+(size_t)sA*N+row0+x;
			a[2+2*k] = fi+(size_t)sB*N+((size_t)zq[cz[k] == 0 ? 0 : cz[k] > 0 ? 1 : 2]*Ny+yq[cy[k] == 0 ? 0 : cy[k] > 0 ? 1 : 2])*Nx+(cx[k] ? xp : x);
		}
		float f[19];
#pragma unroll
		for (int i = 0; i < 19; i++) f[i] = (SYNC == 2 && pass == 1u) ? __hip_atomic_load(a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *a[i];
		float sum = f[0];
#pragma unroll
		for (int i = 1; i < 19; i++) sum += f[i];
#pragma unroll
		for (int i = 0; i < 19; i++) f[i] = f[i]*s+sum*1e-3f+(float)i*1e-4f;
		if (SYNC >= 2 && pass == 0u && passes == 2u) {
			__hip_atomic_store(a[0], f[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
			for (int k = 0; k < 9; k++) { __hip_atomic_store(a[1+2*k], f[2+2*k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(a[2+2*k], f[1+2*k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		} else {
			*a[0] = f[0];
#pragma unroll
			for (int k = 0; k < 9; k++) { *a[1+2*k] = f[2+2*k]; *a[2+2*k] = f[1+2*k]; }
		}
	}
	if (SYNC != 0 && passes == 2u && pass == 0u) { // pass 1 announces the workgroup's rows
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave's stores have been acknowledged before the workgroup signals
		__syncthreads();
		if (threadIdx.x == 0u) { // two levels: a counter per plane and shard (64 bytes apart), whose last arriver reports to the plane's word
			const unsigned sub = in_plane%SUB, want = per_plane/SUB+(sub < per_plane%SUB ? 1u : 0u);
			unsigned* p = done+Nz+(z*SUB+sub)*16u;
			const unsigned old = SYNC == 1 ? __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (old+1u == want) __hip_atomic_fetch_add(done+z, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
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

int main(int argc, char** argv) {
	const unsigned n = argc > 1 ? atoi(argv[1]) : 512;
	const unsigned bpc = argc > 2 ? atoi(argv[2]) : 3;
	const size_t N = (size_t)n*n*n;
	float* fi; unsigned* done; unsigned* error;
	CHECK(hipMalloc(&fi, N*19*sizeof(float)));
	const size_t done_bytes = ((size_t)n+(size_t)n*SUB*16u)*sizeof(unsigned); // plane words, then the shard counters
	CHECK(hipMalloc(&done, done_bytes));
	CHECK(hipMalloc(&error, 4));
	CHECK(hipMemset(error, 0, 4));
	const unsigned b = 256u > n ? n : 256u, xchunks = (n+b-1)/b, per_plane = xchunks*n;
	const unsigned lds = bpc ? (160u*1024u/bpc)&~255u : 0u;
	const double bytes = 152.0*N;
	auto plain = [&](unsigned odd0) {
		hipLaunchKernelGGL(k_skew<0>, dim3(per_plane*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, 0u, 1u, odd0, 0.999f, done, error);
		hipLaunchKernelGGL(k_skew<0>, dim3(per_plane*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, 0u, 1u, odd0^1u, 0.999f, done, error);
	};
	auto skewed = [&](int sync, unsigned lag) {
		if (sync) CHECK(hipMemsetAsync(done, 0, done_bytes, 0));
		if (sync == 0) hipLaunchKernelGGL(k_skew<0>, dim3(per_plane*2u*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, lag, 2u, 0u, 0.999f, done, error);
		if (sync == 1) hipLaunchKernelGGL(k_skew<1>, dim3(per_plane*2u*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, lag, 2u, 0u, 0.999f, done, error);
		if (sync == 3) hipLaunchKernelGGL(k_skew<3>, dim3(per_plane*2u*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, lag, 2u, 0u, 0.999f, done, error);
		if (sync == 2) hipLaunchKernelGGL(k_skew<2>, dim3(per_plane*2u*n), dim3(b), lds, 0, fi, N, n, n, n, xchunks, lag, 2u, 0u, 0.999f, done, error);
	};
	auto fill = [&]() { // a reproducible non-trivial start
		std::vector<float> h((size_t)n*n);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i*2654435761u)>>20)*1e-4f;
		for (size_t off = 0; off < N*19; off += h.size()) CHECK(hipMemcpy(fi+off, h.data(), h.size()*sizeof(float), hipMemcpyHostToDevice));
	};
	auto checksum = [&]() -> unsigned long long { // of a sample of planes of every stream (the full lattice would take long on the host)
		std::vector<unsigned> h((size_t)n*n);
		unsigned long long c = 1469598103934665603ull;
		for (unsigned i = 0; i < 19; i++) for (unsigned z = 0; z < n; z += (z < 4u || z+5u >= n) ? 1u : 13u) {
			CHECK(hipMemcpy(h.data(), fi+(size_t)i*N+(size_t)z*n*n, h.size()*4, hipMemcpyDeviceToHost));
			for (unsigned v : h) { c ^= v; c *= 1099511628211ull; }
		}
		return c;
	};
	fill(); plain(0u); plain(0u); CHECK(hipDeviceSynchronize());
	const unsigned long long want = checksum();
	{
		const float ms = time_ms([&]() { plain(0u); }, 8)/2;
		printf("n=%u blocks/CU<=%u  plain: two launches      %.3f ms per pass  %.0f GB/s (%.1f%% of 8 TB/s)\n", n, bpc, ms, bytes/ms/1e6, bytes/ms/1e6/80); fflush(stdout);
	}
	for (int sync : {0, 2}) for (unsigned lag : {4u, 5u, 6u, 7u}) {
		if (lag+4u >= n) break;
		const char* verdict = "";
		if (sync) {
			fill(); skewed(sync, lag); skewed(sync, lag); CHECK(hipDeviceSynchronize());
			verdict = checksum() == want ? "  bit-identical to plain" : "  DIFFERS from plain";
		}
		const float ms = time_ms([&]() { skewed(sync, lag); }, 8)/2;
		unsigned err = 0; CHECK(hipMemcpy(&err, error, 4, hipMemcpyDeviceToHost));
		printf("n=%u blocks/CU<=%u  sync=%d lag=%2u (window %4.0f MB) %.3f ms per pass  %.0f GB/s-equivalent (%.1f%% of 8 TB/s)%s%s\n", n, bpc, sync, lag,
			(lag+4.0)*19.0*n*n*4/1e6, ms, bytes/ms/1e6, bytes/ms/1e6/80, verdict, err ? "  WAIT TIMED OUT" : ""); fflush(stdout);
		if (err) return 1;
	}
	return 0;
}
