// Microbenchmark (round 5): do LDS reads overlap with independent VALU work of the SAME wave, and does the ORDER in which a wave issues them
// matter?  512 threads (8 waves, 2 per SIMD) per CU; per round every thread issues 16 ds_read_b64 (the exchange pattern of the 8192-point
// kernels) and 96 independent v_pk_fma_f32 (a radix-8 pass and a half), then s_waitcnt lgkmcnt(0) + s_barrier.
//   0: reads only   1: VALU only   2: all 16 reads first, then the VALU work   3: one read per 6 VALU operations, interleaved
//   4: 16 ds_write_b64 only   5: writes first, then VALU   6: VALU with one write per 6 operations interleaved
// hipcc --offload-arch=gfx950 -O3 lds_valu_overlap.hip -o lds_valu_overlap && ./lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

#define FMA6(a, b, c, d, e, f) asm volatile( \
	"v_pk_fma_f32 %0, %0, %6, %7\n v_pk_fma_f32 %1, %1, %6, %7\n v_pk_fma_f32 %2, %2, %6, %7\n" \
	"v_pk_fma_f32 %3, %3, %6, %7\n v_pk_fma_f32 %4, %4, %6, %7\n v_pk_fma_f32 %5, %5, %6, %7" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(cc), "v"(dd))
#define RD(m) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[m]) : "v"(addr), "n"(4096 * (m)) : "memory")
#define WR(m) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(y[m]), "n"(4096 * (m)) : "memory")

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int th = threadIdx.x;
	const unsigned addr = 8u * th;
	v2f x[16], y[16];
	v2f a = {1.0f + th, 2.0f}, b = {3.0f, 4.0f + th}, c = {5.0f, 6.0f}, d = {7.0f, 8.0f}, e = {9.0f, 1.0f}, f = {2.0f, 3.0f};
	const v2f cc = {1.0000001f, 0.9999999f}, dd = {1e-9f, -1e-9f};
	for (int m = 0; m < 16; m++) { x[m] = v2f{0.0f, 0.0f}; y[m] = v2f{(float)th, (float)m}; }
	for (int i = th; i < 16384; i += 512) reinterpret_cast<float *>(smem)[i] = (float)i;
	__syncthreads();
	for (int i = 0; i < iters; i++) {
		if (KIND == 0) { RD(0); RD(1); RD(2); RD(3); RD(4); RD(5); RD(6); RD(7); RD(8); RD(9); RD(10); RD(11); RD(12); RD(13); RD(14); RD(15); }
		if (KIND == 1) { for (int r = 0; r < 16; r++) FMA6(a, b, c, d, e, f); }
		if (KIND == 2) { RD(0); RD(1); RD(2); RD(3); RD(4); RD(5); RD(6); RD(7); RD(8); RD(9); RD(10); RD(11); RD(12); RD(13); RD(14); RD(15);
		                 for (int r = 0; r < 16; r++) FMA6(a, b, c, d, e, f); }
		if (KIND == 3) {
#define STEP(m) RD(m); FMA6(a, b, c, d, e, f);
			STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
		}
		if (KIND == 4) { WR(0); WR(1); WR(2); WR(3); WR(4); WR(5); WR(6); WR(7); WR(8); WR(9); WR(10); WR(11); WR(12); WR(13); WR(14); WR(15); }
		if (KIND == 5) { WR(0); WR(1); WR(2); WR(3); WR(4); WR(5); WR(6); WR(7); WR(8); WR(9); WR(10); WR(11); WR(12); WR(13); WR(14); WR(15);
		                 for (int r = 0; r < 16; r++) FMA6(a, b, c, d, e, f); }
		if (KIND == 6) {
#define STEP(m) FMA6(a, b, c, d, e, f); WR(m);
			STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
		}
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
	float r = a.x + b.y + c.x + d.y + e.x + f.y;
	for (int m = 0; m < 16; m++) r += x[m].x + x[m].y;
	if (r == 12345.678f) out[0] = r;
}

int main()
{
	float *dmem; (void)hipMalloc(&dmem, 64 * sizeof(float));
	const char *names[] = {"16 reads", "96 pk_fma", "16 reads, then 96 pk_fma", "1 read : 6 pk_fma interleaved", "16 writes", "16 writes, then 96 pk_fma", "6 pk_fma : 1 write interleaved"};
	for (int kind = 0; kind < 7; kind++) {
		const int iters = 2000;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		auto launch = [&]() {
			switch (kind) {
			case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 65536, 0, dmem, iters); break;
			}
		};
		launch(); (void)hipDeviceSynchronize();
		(void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		printf("%-34s %7.1f ns per round per CU\n", names[kind], ms * 1e6 / iters);
	}
	return 0;
}
