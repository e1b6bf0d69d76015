// Microbenchmark: does the VGPR bank of the two 64-bit source operands of v_pk_add_f32 / v_pk_mul_f32 matter on gfx950?
// Explicit registers: destination v[20:27], sources from bank-aligned pairs.
// hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank && ./vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
	long long t0 = clock64();
	for (int it = 0; it < iters; it++) {
		if (KIND == 0) {	// sources v[4:5], v[8:9]: both pairs start in bank 0
			REP64(asm volatile("v_pk_add_f32 v[20:21], v[4:5], v[8:9]\n v_pk_add_f32 v[22:23], v[4:5], v[8:9]\n"
			                   "v_pk_add_f32 v[24:25], v[4:5], v[8:9]\n v_pk_add_f32 v[26:27], v[4:5], v[8:9]" ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
		} else if (KIND == 1) {	// v[4:5] (banks 0,1), v[10:11] (banks 2,3)
			REP64(asm volatile("v_pk_add_f32 v[20:21], v[4:5], v[10:11]\n v_pk_add_f32 v[22:23], v[4:5], v[10:11]\n"
			                   "v_pk_add_f32 v[24:25], v[4:5], v[10:11]\n v_pk_add_f32 v[26:27], v[4:5], v[10:11]" ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
		} else if (KIND == 2) {	// same register twice
			REP64(asm volatile("v_pk_add_f32 v[20:21], v[4:5], v[4:5]\n v_pk_add_f32 v[22:23], v[4:5], v[4:5]\n"
			                   "v_pk_add_f32 v[24:25], v[4:5], v[4:5]\n v_pk_add_f32 v[26:27], v[4:5], v[4:5]" ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
		} else if (KIND == 3) {	// with op_sel / neg modifiers, different banks
			REP64(asm volatile("v_pk_add_f32 v[20:21], v[4:5], v[10:11] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 v[22:23], v[4:5], v[10:11] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
			                   "v_pk_mul_f32 v[24:25], v[4:5], v[10:11] op_sel_hi:[1,0]\n v_pk_mul_f32 v[26:27], v[4:5], v[10:11] op_sel:[1,1] op_sel_hi:[0,1]" ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
		} else if (KIND == 4) {	// scalar adds, sources same bank (v4, v8) vs
			REP64(asm volatile("v_add_f32 v20, v4, v8\n v_add_f32 v21, v4, v8\n v_add_f32 v22, v4, v8\n v_add_f32 v23, v4, v8" ::: "v20","v21","v22","v23");)
		} else if (KIND == 5) {	// different banks (v4, v9)
			REP64(asm volatile("v_add_f32 v20, v4, v9\n v_add_f32 v21, v4, v9\n v_add_f32 v22, v4, v9\n v_add_f32 v23, v4, v9" ::: "v20","v21","v22","v23");)
		} else if (KIND == 6) {	// dependent pk chain through different registers: dst of one is src of next
			REP64(asm volatile("v_pk_add_f32 v[20:21], v[26:27], v[10:11]\n v_pk_add_f32 v[22:23], v[20:21], v[10:11]\n"
			                   "v_pk_add_f32 v[24:25], v[22:23], v[10:11]\n v_pk_add_f32 v[26:27], v[24:25], v[10:11]" ::: "v20","v21","v22","v23","v24","v25","v26","v27");)
		}
	}
	long long t1 = clock64();
	if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * 256);
}

template <int KIND> static void run(float *d, const char *name)
{
	for (int wps = 1; wps <= 2; wps++) {
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		const int iters = 2000;
		hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
		hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("%-44s waves/SIMD %d: %.2f ns per instr per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms * 1e6 / (iters * 256.0 * wps), ms * 1e6 / (iters * 256.0 * wps) * 2.4);
	}
}

int main()
{
	float *d; hipMalloc(&d, 64 * sizeof(float));
	run<0>(d, "v_pk_add_f32 srcs v[4:5], v[8:9] (same banks)");
	run<1>(d, "v_pk_add_f32 srcs v[4:5], v[10:11] (other banks)");
	run<2>(d, "v_pk_add_f32 srcs v[4:5], v[4:5]");
	run<3>(d, "pk add/mul with op_sel/neg modifiers");
	run<4>(d, "v_add_f32 srcs v4, v8 (same bank)");
	run<5>(d, "v_add_f32 srcs v4, v9 (other bank)");
	run<6>(d, "v_pk_add_f32 dependent chain");
	return 0;
}
