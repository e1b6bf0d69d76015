// Microbenchmark (round 5): does the issue cost of v_pk_fma_f32 / v_pk_mul_f32 / v_fma_f32 on gfx950 depend on WHICH registers the sources sit in
// (VGPR bank = register number mod 4)?  Explicit physical registers, 8 independent destinations, 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank && ./vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define STR2(x) #x
#define STR(x) STR2(x)
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"
// 8 destinations v[40:41] .. v[54:55]; sources named per case
#define PKFMA8(A, B, C) \
	"v_pk_fma_f32 v[40:41], " A ", " B ", " C "\n v_pk_fma_f32 v[42:43], " A ", " B ", " C "\n v_pk_fma_f32 v[44:45], " A ", " B ", " C "\n v_pk_fma_f32 v[46:47], " A ", " B ", " C "\n" \
	"v_pk_fma_f32 v[48:49], " A ", " B ", " C "\n v_pk_fma_f32 v[50:51], " A ", " B ", " C "\n v_pk_fma_f32 v[52:53], " A ", " B ", " C "\n v_pk_fma_f32 v[54:55], " A ", " B ", " C "\n"
#define PKMUL8(A, B) \
	"v_pk_mul_f32 v[40:41], " A ", " B "\n v_pk_mul_f32 v[42:43], " A ", " B "\n v_pk_mul_f32 v[44:45], " A ", " B "\n v_pk_mul_f32 v[46:47], " A ", " B "\n" \
	"v_pk_mul_f32 v[48:49], " A ", " B "\n v_pk_mul_f32 v[50:51], " A ", " B "\n v_pk_mul_f32 v[52:53], " A ", " B "\n v_pk_mul_f32 v[54:55], " A ", " B "\n"
#define FMA8(A, B, C) \
	"v_fma_f32 v40, " A ", " B ", " C "\n v_fma_f32 v41, " A ", " B ", " C "\n v_fma_f32 v42, " A ", " B ", " C "\n v_fma_f32 v43, " A ", " B ", " C "\n" \
	"v_fma_f32 v44, " A ", " B ", " C "\n v_fma_f32 v45, " A ", " B ", " C "\n v_fma_f32 v46, " A ", " B ", " C "\n v_fma_f32 v47, " A ", " B ", " C "\n"
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
	asm volatile("v_mov_b32 v60, 1.0\n v_mov_b32 v61, 1.0\n v_mov_b32 v62, 1.0\n v_mov_b32 v63, 1.0\n v_mov_b32 v64, 1.0\n v_mov_b32 v65, 1.0\n v_mov_b32 v66, 1.0\n v_mov_b32 v67, 1.0\n"
	             "v_mov_b32 v68, 1.0\n v_mov_b32 v69, 1.0\n v_mov_b32 v70, 1.0\n v_mov_b32 v71, 1.0\n v_mov_b32 v72, 1.0\n v_mov_b32 v73, 1.0\n v_mov_b32 v74, 1.0\n v_mov_b32 v75, 1.0\n" ::: CLOB);
	for (int i = 0; i < iters; i++) {
		if (KIND == 0)      asm volatile(REP8(PKFMA8("v[60:61]", "v[62:63]", "v[64:65]")) ::: CLOB);	// banks (0,1) (2,3) (0,1)
		else if (KIND == 1) asm volatile(REP8(PKFMA8("v[60:61]", "v[64:65]", "v[68:69]")) ::: CLOB);	// (0,1) x 3
		else if (KIND == 2) asm volatile(REP8(PKFMA8("v[60:61]", "v[60:61]", "v[62:63]")) ::: CLOB);	// src0 == src1
		else if (KIND == 3) asm volatile(REP8(PKFMA8("v[60:61]", "v[60:61]", "v[60:61]")) ::: CLOB);	// all the same
		else if (KIND == 4) asm volatile(REP8(PKFMA8("v[60:61]", "s[20:21]", "v[62:63]")) ::: CLOB);	// one SGPR pair
		else if (KIND == 5) asm volatile(REP8(PKMUL8("v[60:61]", "v[62:63]")) ::: CLOB);
		else if (KIND == 6) asm volatile(REP8(PKMUL8("v[60:61]", "v[64:65]")) ::: CLOB);
		else if (KIND == 7) asm volatile(REP8(PKMUL8("v[60:61]", "v[60:61]")) ::: CLOB);
		else if (KIND == 8) asm volatile(REP8(FMA8("v60", "v61", "v62")) ::: CLOB);			// banks 0 1 2
		else if (KIND == 9) asm volatile(REP8(FMA8("v60", "v64", "v68")) ::: CLOB);			// 0 0 0
		else if (KIND == 10) asm volatile(REP8(PKFMA8("v[60:61]", "v[62:63]", "v[66:67]")) ::: CLOB);	// (0,1) (2,3) (2,3)
		else if (KIND == 11) asm volatile(REP8(PKFMA8("v[60:61]", "v[62:63]", "v[64:65]") "s_nop 0\n") ::: CLOB);
	}
	if (out[0] == 12345.0f) asm volatile("global_store_dword %0, v40, off" :: "v"(out) : "memory");
}

int main()
{
	float *d; (void)hipMalloc(&d, 64 * sizeof(float)); (void)hipMemset(d, 0, 64 * sizeof(float));
	const char *names[] = {"pk_fma banks 01 23 01", "pk_fma banks 01 01 01", "pk_fma src0 == src1", "pk_fma all same", "pk_fma one SGPR src", "pk_mul banks 01 23", "pk_mul banks 01 01",
	                       "pk_mul same reg", "v_fma banks 0 1 2", "v_fma banks 0 0 0", "pk_fma banks 01 23 23", "pk_fma + s_nop per 8"};
	for (int wpb = 1; wpb <= 4; wpb *= 2) {
		for (int kind = 0; kind < 12; kind++) {
			hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
			dim3 grid(256 * wpb), block(256);
			const int iters = 2000;
			auto launch = [&]() {
				switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, grid, block, 0, 0, d, iters); break;
				C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
				}
			};
			launch(); (void)hipDeviceSynchronize();
			(void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1);
			const double ns = ms * 1e6 / ((double)wpb * iters * 64);
			printf("waves/SIMD %d  %-24s %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", wpb, names[kind], ns, ns * 2.4);
		}
	}
	return 0;
}
