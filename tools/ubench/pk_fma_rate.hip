// Microbenchmark (round 5): what v_pk_fma_f32 costs on gfx950 next to v_pk_mul_f32 / v_pk_add_f32 -- issue cost with independent
// chains, latency of a dependent chain, and the butterfly of the long FFT plans (bf(): fma -> fma -> fma) eight at a time.
// hipcc --offload-arch=gfx950 -O3 pk_fma_rate.hip -o pk_fma_rate && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
	float a0 = seed + threadIdx.x;
	v2f p0 = {a0, a0 + 1}, p1 = {a0 + 2, a0 + 3}, p2 = {a0 + 4, a0 + 5}, p3 = {a0 + 6, a0 + 7}, p4 = {a0 + 1, a0}, p5 = {a0 + 3, a0 + 2}, p6 = {a0 + 5, a0 + 4}, p7 = {a0 + 7, a0 + 6};
	v2f q0 = p7, q1 = p6, q2 = p5, q3 = p4, q4 = p3, q5 = p2, q6 = p1, q7 = p0;
	const v2f cc = {1.0000001f, 0.9999999f}, dd = {1e-9f, -1e-9f}, two = {2.0f, 2.0f};
	int n_inst = 128;
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) {	// pk_mul, 8 independent chains
			REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
			                    "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
			                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc));)
		} else if (KIND == 1) {	// pk_fma, 3 VGPR sources, 8 independent chains
			REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
			                    "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
			                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc), "v"(dd));)
		} else if (KIND == 2) {	// pk_fma with op_sel / neg modifiers (the butterfly's first step)
			REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
			                    "v_pk_fma_f32 %2, %2, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
			                    "v_pk_fma_f32 %4, %4, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
			                    "v_pk_fma_f32 %6, %6, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
			                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc), "v"(dd));)
		} else if (KIND == 3) {	// pk_fma, ONE dependent chain (latency)
			REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
			                    "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
			                    : "+v"(p0) : "v"(cc), "v"(dd));)
		} else if (KIND == 4) {	// pk_mul, one dependent chain
			REP16(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n"
			                    "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n"
			                    : "+v"(p0) : "v"(cc));)
		} else if (KIND == 5) {	// the butterfly bf(a, b, t): three dependent pk_fma; 8 butterflies interleaved step by step (24 instr)
			n_inst = 24 * 16;
			REP16(asm volatile(
				"v_pk_fma_f32 %8, %8, %16, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %9, %9, %16, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
				"v_pk_fma_f32 %10, %10, %16, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %11, %11, %16, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
				"v_pk_fma_f32 %12, %12, %16, %4 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %13, %13, %16, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
				"v_pk_fma_f32 %14, %14, %16, %6 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %15, %15, %16, %7 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n"
				"v_pk_fma_f32 %8, %0, %16, %8 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %9, %1, %16, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %10, %2, %16, %10 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %11, %3, %16, %11 op_sel_hi:[1,0,1]\n"
				"v_pk_fma_f32 %12, %4, %16, %12 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %13, %5, %16, %13 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %14, %6, %16, %14 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %15, %7, %16, %15 op_sel_hi:[1,0,1]\n"
				"v_pk_fma_f32 %0, %0, %17, %8 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %1, %1, %17, %9 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %2, %2, %17, %10 neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
				"v_pk_fma_f32 %3, %3, %17, %11 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %4, %4, %17, %12 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %5, %5, %17, %13 neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
				"v_pk_fma_f32 %6, %6, %17, %14 neg_lo:[0,0,1] neg_hi:[0,0,1]\n v_pk_fma_f32 %7, %7, %17, %15 neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
				: "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
				: "v"(cc), "v"(two));)
		} else if (KIND == 6) {	// the old form of the same work: complex product (mul, mul, add) + butterfly (add, sub) = 5 per pair, 8 pairs (40 instr)
			n_inst = 40 * 16;
			REP16(asm volatile(
				"v_pk_mul_f32 %8, %0, %16 op_sel_hi:[1,0]\n v_pk_mul_f32 %9, %1, %16 op_sel_hi:[1,0]\n v_pk_mul_f32 %10, %2, %16 op_sel_hi:[1,0]\n v_pk_mul_f32 %11, %3, %16 op_sel_hi:[1,0]\n"
				"v_pk_mul_f32 %12, %4, %16 op_sel_hi:[1,0]\n v_pk_mul_f32 %13, %5, %16 op_sel_hi:[1,0]\n v_pk_mul_f32 %14, %6, %16 op_sel_hi:[1,0]\n v_pk_mul_f32 %15, %7, %16 op_sel_hi:[1,0]\n"
				"v_pk_mul_f32 %0, %0, %16 op_sel:[1,1] op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %16 op_sel:[1,1] op_sel_hi:[0,1]\n v_pk_mul_f32 %2, %2, %16 op_sel:[1,1] op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %16 op_sel:[1,1] op_sel_hi:[0,1]\n"
				"v_pk_mul_f32 %4, %4, %16 op_sel:[1,1] op_sel_hi:[0,1]\n v_pk_mul_f32 %5, %5, %16 op_sel:[1,1] op_sel_hi:[0,1]\n v_pk_mul_f32 %6, %6, %16 op_sel:[1,1] op_sel_hi:[0,1]\n v_pk_mul_f32 %7, %7, %16 op_sel:[1,1] op_sel_hi:[0,1]\n"
				"v_pk_add_f32 %8, %8, %0 neg_lo:[0,1]\n v_pk_add_f32 %9, %9, %1 neg_lo:[0,1]\n v_pk_add_f32 %10, %10, %2 neg_lo:[0,1]\n v_pk_add_f32 %11, %11, %3 neg_lo:[0,1]\n"
				"v_pk_add_f32 %12, %12, %4 neg_lo:[0,1]\n v_pk_add_f32 %13, %13, %5 neg_lo:[0,1]\n v_pk_add_f32 %14, %14, %6 neg_lo:[0,1]\n v_pk_add_f32 %15, %15, %7 neg_lo:[0,1]\n"
				"v_pk_add_f32 %0, %17, %8\n v_pk_add_f32 %1, %17, %9\n v_pk_add_f32 %2, %17, %10\n v_pk_add_f32 %3, %17, %11\n v_pk_add_f32 %4, %17, %12\n v_pk_add_f32 %5, %17, %13\n v_pk_add_f32 %6, %17, %14\n v_pk_add_f32 %7, %17, %15\n"
				"v_pk_add_f32 %8, %17, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %9, %17, %9 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %10, %17, %10 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %11, %17, %11 neg_lo:[0,1] neg_hi:[0,1]\n"
				"v_pk_add_f32 %12, %17, %12 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %13, %17, %13 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %14, %17, %14 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %15, %17, %15 neg_lo:[0,1] neg_hi:[0,1]\n"
				: "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
				: "v"(cc), "v"(two));)
		} else if (KIND == 7) {	// scalar fma, 3 VGPR sources, 8 chains (two of these do the work of one pk_fma)
			REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
			                    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
			                    : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(cc.x), "v"(dd.x));)
		}
	}
	float r = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y;
	if (r == 12345.678f) out[0] = r;
	if (threadIdx.x == 0 && blockIdx.x == 0) out[1 + KIND] = (float)n_inst;
}

int main()
{
	float *d; hipMalloc(&d, 64 * sizeof(float)); hipMemset(d, 0, 64 * sizeof(float));
	const char *names[] = {"pk_mul x8 chains", "pk_fma x8 chains", "pk_fma(mods) x8", "pk_fma 1 chain", "pk_mul 1 chain", "bf x8 (24 pk_fma)", "cmul+dft2 x8 (40 pk)", "v_fma_f32 x8 chains"};
	for (int wpb = 1; wpb <= 2; wpb *= 2) {
		for (int kind = 0; kind < 8; kind++) {
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			dim3 grid(256 * wpb), block(256);
			const int iters = 1000;
			auto launch = [&]() {
				switch (kind) {
				case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 4: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 5: hipLaunchKernelGGL(k<5>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 6: hipLaunchKernelGGL(k<6>, grid, block, 0, 0, d, iters, 1.0f); break;
				case 7: hipLaunchKernelGGL(k<7>, grid, block, 0, 0, d, iters, 1.0f); break;
				}
			};
			launch(); hipDeviceSynchronize();
			hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
			const double n_inst = h[1 + kind];
			double ns_per_inst = ms * 1e6 / ((double)wpb * iters * n_inst);
			printf("waves/SIMD %d  %-22s  %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz); per 8 butterflies: %.0f cyc\n",
			       wpb, names[kind], ns_per_inst, ns_per_inst * 2.4, (kind == 5 ? 24 : kind == 6 ? 40 : 0) * ns_per_inst * 2.4);
		}
	}
	return 0;
}
