// Microbenchmark: cost of DEPENDENT VALU instructions on gfx950 -- how many independent instructions a wave
// needs between a producer and its consumer to issue back to back.  One wave per SIMD (and 2), chains of
// 1 / 2 / 4 / 8 independent dependency chains, for v_fma_f32, v_pk_add_f32, v_log_f32 -> v_fma_f32, v_rndne.
// hipcc --offload-arch=gfx950 -O3 valu_dep.hip -o valu_dep && ./valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND, int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
	float a[8]; v2f p[8];
	for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; p[i] = v2f{a[i], a[i] + 1}; }
	const float c = 1.0000001f; const v2f cc = {c, c};
	long long t0 = clock64();
	for (int it = 0; it < iters; it++) {
		if (KIND == 0) {		// v_fma_f32 chains
			if (CH == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(c));) }
			if (CH == 2) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(a[0]), "+v"(a[1]) : "v"(c));) }
			if (CH == 4) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c));) }
		} else if (KIND == 1) {	// v_pk_add_f32 chains
			if (CH == 1) { REP64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(cc));) }
			if (CH == 2) { REP64(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(p[0]), "+v"(p[1]) : "v"(cc));) }
			if (CH == 4) { REP64(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(cc));) }
		} else if (KIND == 2) {	// v_log_f32 -> v_fma_f32 (trans result consumed at once)
			if (CH == 1) { REP64(asm volatile("v_log_f32 %0, %0\n v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(c));) }
			if (CH == 2) { REP64(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(a[0]), "+v"(a[1]) : "v"(c));) }
			if (CH == 4) { REP64(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c));) }
		} else if (KIND == 3) {	// v_mul_f32 chains (the "2.5-cycle" class)
			if (CH == 1) { REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));) }
			if (CH == 2) { REP64(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a[0]), "+v"(a[1]) : "v"(c));) }
			if (CH == 4) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(c));) }
		}
	}
	long long t1 = clock64();
	float r = 0; for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y;
	if (r == 12345.678f) out[0] = r;
	const int per = (KIND == 2) ? 2 * CH * 64 : CH * 64;
	if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0) / (float)((long long)iters * per);
}

template <int KIND, int CH>
static void run(float *d, const char *name, int wps)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 2000;
	const int per = (KIND == 2) ? 2 * CH * 64 : CH * 64;
	hipLaunchKernelGGL((k<KIND, CH>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f); hipDeviceSynchronize();
	hipEventRecord(e0); hipLaunchKernelGGL((k<KIND, CH>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	float h[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	printf("waves/SIMD %d  %-22s chains %d : %.2f ns per instr per wave  (s_memtime units/instr %.2f)\n",
	       wps, name, CH, ms * 1e6 / ((double)iters * per), h[1]);
}

int main()
{
	float *d; hipMalloc(&d, 64 * sizeof(float)); hipMemset(d, 0, 64 * sizeof(float));
	for (int wps = 1; wps <= 2; wps++) {
		run<0, 1>(d, "v_fma_f32", wps); run<0, 2>(d, "v_fma_f32", wps); run<0, 4>(d, "v_fma_f32", wps);
		run<1, 1>(d, "v_pk_add_f32", wps); run<1, 2>(d, "v_pk_add_f32", wps); run<1, 4>(d, "v_pk_add_f32", wps);
		run<2, 1>(d, "v_log_f32->v_fma_f32", wps); run<2, 2>(d, "v_log_f32->v_fma_f32", wps); run<2, 4>(d, "v_log_f32->v_fma_f32", wps);
		run<3, 1>(d, "v_mul_f32", wps); run<3, 2>(d, "v_mul_f32", wps); run<3, 4>(d, "v_mul_f32", wps);
	}
	return 0;
}
