// Microbenchmark: issue cost of scalar vs packed fp32 VALU ops on gfx950 (cycles per wave-instruction per SIMD).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
	const float c = 1.0000001f;
	const v2f cc = {c, c};
	long long t0 = clock64();
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) {	// scalar mul, 8 independent chains x 16
			REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
			                    "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
		} else if (KIND == 1) {	// packed mul
			REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
			                    "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
			                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
		} else if (KIND == 2) {	// v_mov
			REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
			                    "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
		} else if (KIND == 3) {	// v_log_f32
			REP16(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
			                    "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
		} else if (KIND == 4) {	// scalar fma
			REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
			                    "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
		} else if (KIND == 5) {	// packed add with modifiers
			REP16(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
			                    "v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
			                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
		} else if (KIND == 6) {	// v_cndmask + v_cmp
			REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
			                    "v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
			                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
		}
	}
	long long t1 = clock64();
	float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
	if (r == 12345.678f) out[0] = r;
	if (threadIdx.x == 0 && blockIdx.x == 0) out[1 + KIND] = (float)(t1 - t0) / (float)(iters * 128);
}

int main()
{
	float *d; hipMalloc(&d, 64 * sizeof(float)); hipMemset(d, 0, 64 * sizeof(float));
	const char *names[] = {"v_mul_f32", "v_pk_mul_f32", "v_mov_b32", "v_log_f32", "v_fma_f32", "v_pk_add_f32(mod)", "v_cmp+v_cndmask"};
	for (int wpb = 1; wpb <= 4; wpb *= 2) {	// waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
		for (int kind = 0; kind < 7; kind++) {
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			dim3 grid(256 * wpb), block(256);
			const int iters = 2000;
			auto launch = [&]() {
				switch (kind) {
				case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 6: hipLaunchKernelGGL(k<6>, grid, block, 0, 0, d, iters, 1.0f); break;
				}
			};
			launch(); hipDeviceSynchronize();
			hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
			// instructions per SIMD = wpb waves * iters * 128
			double ns_per_inst = ms * 1e6 / ((double)wpb * iters * 128);
			printf("waves/SIMD %d  %-20s  %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)   s_memtime/instr (one wave) %.2f\n",
			       wpb, names[kind], ns_per_inst, ns_per_inst * 2.4, h[1 + kind]);
		}
	}
	return 0;
}
