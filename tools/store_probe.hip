// Diagnostic: what do the x-shifted stores of initialize cost?  19 streams of 512^3 floats, one workgroup per 256 cells of a row.
// mode 0: every lane stores 19 dwords at its own x; mode 1: 5 streams at x+1 and 5 at x-1 (periodic), like store_f;
// mode 2: the same values but every wave writes aligned segments (the shifted element comes from the neighbouring lane,
// the seam lane writes a dummy); mode 3: mode 1 plus reading 17 bytes per cell (rho, u, flags).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(s_)); return 1; } } while (0)
template <int MODE> __global__ __launch_bounds__(256) void k(float* fi, const float* rho, const unsigned char* flags, unsigned long long S, unsigned Nx, unsigned Ny) {
	const unsigned chunks = Nx/256u, chunk = blockIdx.x%chunks, y = blockIdx.x/chunks, z = blockIdx.y;
	const unsigned x = chunk*256u+threadIdx.x;
	const unsigned long long row0 = ((unsigned long long)z*Ny+y)*Nx;
	float v = (float)x;
	if (MODE == 3) v = rho[row0+x]+rho[S+row0+x]+rho[2*S+row0+x]+rho[3*S+row0+x]+(float)flags[row0+x];
	const unsigned xp = x+1u == Nx ? 0u : x+1u, xm = x == 0u ? Nx-1u : x-1u;
#pragma unroll
	for (int i = 0; i < 19; i++) {
		float* p = fi+(unsigned long long)i*S+row0;
		const float val = v+(float)i;
		if (MODE == 0 || i < 9) p[x] = val;
		else if (MODE == 1 || MODE == 3) p[i < 14 ? xp : xm] = val;
		else { const float t = i < 14 ? __shfl_up(val, 1) : __shfl_down(val, 1); p[x] = t; }
	}
}
int main(int argc, char** argv) {
	const unsigned n = argc > 1 ? atoi(argv[1]) : 512;
	const unsigned long long N = (unsigned long long)n*n*n;
	float* fi; float* rho; unsigned char* flags;
	CHECK(hipMalloc(&fi, 19*N*4)); CHECK(hipMalloc(&rho, 4*N*4)); CHECK(hipMalloc(&flags, N));
	CHECK(hipMemset(rho, 0, 4*N*4)); CHECK(hipMemset(flags, 0, N));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int mode = 0; mode < 4; mode++) {
		float best = 1e30f;
		for (int rep = 0; rep < 6; rep++) {
			CHECK(hipEventRecord(e0));
			const dim3 grid(n/256*n, n), block(256);
			if (mode == 0) k<0><<<grid, block>>>(fi, rho, flags, N, n, n);
			if (mode == 1) k<1><<<grid, block>>>(fi, rho, flags, N, n, n);
			if (mode == 2) k<2><<<grid, block>>>(fi, rho, flags, N, n, n);
			if (mode == 3) k<3><<<grid, block>>>(fi, rho, flags, N, n, n);
			CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (rep > 0 && ms < best) best = ms;
		}
		const double bytes = (double)N*(76+(mode == 3 ? 17 : 0));
		printf("mode %d: %.3f ms  %.0f GB/s (%.1f %% of 8 TB/s)\n", mode, best, bytes/best/1e6, bytes/best/1e6/80);
	}
	return 0;
}
