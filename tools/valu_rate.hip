// Diagnostic: issue rate of scalar-per-lane FP32 (v_fma_f32, v_add_f32) against packed FP32 (v_pk_fma_f32, v_pk_add_f32) on
// the running GPU: 16 independent dependency chains per lane, enough waves to fill every SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/valu_rate.hip -o tools/valu_rate ; run: tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int CHAINS = 16, ITER = 4096;

template <int MODE> __global__ __launch_bounds__(256) void k_rate(float* out, float a, float b) {
	if constexpr (MODE == 0 || MODE == 2) {
		float x[CHAINS];
		for (int c = 0; c < CHAINS; c++) x[c] = (float)(threadIdx.x+c);
		for (int it = 0; it < ITER; it++) {
#pragma unroll
			for (int c = 0; c < CHAINS; c++) x[c] = MODE == 0 ? fmaf(x[c], a, b) : x[c]+a;
		}
		float s = 0.0f;
		for (int c = 0; c < CHAINS; c++) s += x[c];
		if (s == 12345.678f) out[0] = s;
	} else {
		float2v x[CHAINS];
		for (int c = 0; c < CHAINS; c++) { x[c].x = (float)(threadIdx.x+c); x[c].y = (float)(threadIdx.x-c); }
		const float2v av = {a, a+1.0f}, bv = {b, b+1.0f};
		for (int it = 0; it < ITER; it++) {
#pragma unroll
			for (int c = 0; c < CHAINS; c++) x[c] = MODE == 1 ? __builtin_elementwise_fma(x[c], av, bv) : x[c]+av;
		}
		float s = 0.0f;
		for (int c = 0; c < CHAINS; c++) s += x[c].x+x[c].y;
		if (s == 12345.678f) out[0] = s;
	}
}

template <int MODE> static void run(const char* what, float* out, int per_instr) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256*16;
	hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.0f; hipEventElapsedTime(&ms, e0, e1);
	const double wave_instr = (double)blocks*4.0*CHAINS*ITER;
	std::printf("%-14s %8.3f ms  %7.1f G wave-instructions/s  %6.1f TFLOP/s\n", what, ms, wave_instr/ms/1e6, wave_instr*64.0*per_instr/ms/1e9);
}

int main() {
	float* out; hipMalloc(&out, 64);
	run<0>("v_fma_f32", out, 2);
	run<1>("v_pk_fma_f32", out, 4);
	run<2>("v_add_f32", out, 1);
	run<3>("v_pk_add_f32", out, 2);
	hipFree(out);
	return 0;
}
