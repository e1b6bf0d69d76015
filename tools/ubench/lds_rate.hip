// Microbenchmark (round 5): LDS throughput of one CU for the exchange pattern of the 8192-point kernels -- 512 threads (8 waves), every
// thread 16 x ds_write_b64 then 16 x ds_read_b64 of lane-contiguous 8-byte elements (th + 512 m), conflict-free -- in clocks per
// wave-instruction per CU, for reads, writes and both, b64 and b128.
// hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate && ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	v2f *s2 = reinterpret_cast<v2f *>(smem);
	v4f *s4 = reinterpret_cast<v4f *>(smem);
	const int th = threadIdx.x;
	v2f x[16];
	v4f y[8];
	for (int m = 0; m < 16; m++) x[m] = v2f{ (float)th, (float)m };
	for (int m = 0; m < 8; m++) y[m] = v4f{ (float)th, (float)m, 1.0f, 2.0f };
	for (int m = 0; m < 16; m++) s2[th + 512 * m] = x[m];
	__syncthreads();
	long long t0 = clock64();
	for (int i = 0; i < iters; i++) {
		if (KIND == 0 || KIND == 2) {		// 16 x ds_write_b64
#pragma unroll
			for (int m = 0; m < 16; m++) s2[th + 512 * m] = x[m];
		}
		if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		if (KIND == 1 || KIND == 2) {		// 16 x ds_read_b64
#pragma unroll
			for (int m = 0; m < 16; m++) x[m] += s2[th + 512 * m];
		}
		if (KIND == 3) {			// 8 x ds_write_b128
#pragma unroll
			for (int m = 0; m < 8; m++) s4[th + 512 * m] = y[m];
		}
		if (KIND == 4) {			// 8 x ds_read_b128
#pragma unroll
			for (int m = 0; m < 8; m++) y[m] += s4[th + 512 * m];
		}
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
	long long t1 = clock64();
	float r = 0;
	for (int m = 0; m < 16; m++) r += x[m].x + x[m].y;
	for (int m = 0; m < 8; m++) r += y[m].x + y[m].w;
	if (r == 12345.678f) out[0] = r;
	if (th == 0 && blockIdx.x == 0) out[1 + KIND] = (float)(t1 - t0) / (float)iters;
}

int main()
{
	float *d; (void)hipMalloc(&d, 64 * sizeof(float)); (void)hipMemset(d, 0, 64 * sizeof(float));
	const char *names[] = {"16 ds_write_b64 / thread", "16 ds_read_b64 / thread", "16 writes + barrier + 16 reads", "8 ds_write_b128 / thread", "8 ds_read_b128 / thread"};
	const int bytes[] = {65536, 65536, 131072, 65536, 65536};
	for (int kind = 0; kind < 5; kind++) {
		const int iters = 2000;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		auto launch = [&]() {
			switch (kind) {
			case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, d, iters); break;
			case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, d, iters); break;
			case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 65536, 0, d, iters); break;
			case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 65536, 0, d, iters); break;
			case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 65536, 0, d, iters); break;
			}
		};
		launch(); (void)hipDeviceSynchronize();
		(void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		float h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		const double ns = ms * 1e6 / iters;
		printf("%-34s %8.1f ns per round per CU (%6.0f s_memtime cycles): %5.1f B/ns per CU = %5.1f B/clk @2.1 GHz\n",
		       names[kind], ns, h[1 + kind], bytes[kind] / ns, bytes[kind] / ns / 2.1);
	}
	return 0;
}
