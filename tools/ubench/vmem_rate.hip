// Microbenchmark (round 5): what one vector-memory instruction costs a CU on gfx950 -- stores of 2 / 4 / 8 / 16 bytes per lane and loads of 8 / 16,
// 8 waves per CU (one 512-thread work-group), every wave on its own L2-resident region, 16 instructions per wave and round, no other work.
// hipcc --offload-arch=gfx950 -O3 vmem_rate.hip -o vmem_rate && ./vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(char *buf, int iters, uint32_t *sink)
{
	// region per work-group: 64 KiB; per wave 8 KiB; lane stride = access size (dense) or 4 B for the strided short case
	char *base = buf + (size_t)blockIdx.x * 65536 + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 8192;
	const uint32_t lane = threadIdx.x & 63;
	uint32_t v = threadIdx.x;
	u2v v2 = { v, v }; u4v v4 = { v, v, v, v };
	u2v a2 = { 0, 0 }; u4v a4 = { 0, 0, 0, 0 };
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int j = 0; j < 16; j++) {
			if (KIND == 0)      asm volatile("global_store_short %0, %1, %2" :: "v"(lane * 4 + (j & 1) * 2), "v"(v), "s"(base + 256 * (j >> 1)) : "memory");	// strided shorts (the index rows)
			else if (KIND == 1) asm volatile("global_store_short %0, %1, %2" :: "v"(lane * 2), "v"(v), "s"(base + 128 * j) : "memory");				// dense shorts
			else if (KIND == 2) asm volatile("global_store_dword %0, %1, %2" :: "v"(lane * 4), "v"(v), "s"(base + 256 * j) : "memory");
			else if (KIND == 3) asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(lane * 8), "v"(v2), "s"(base + 512 * (j & 7)) : "memory");
			else if (KIND == 4) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(lane * 16), "v"(v4), "s"(base + 1024 * (j & 7)) : "memory");
			else if (KIND == 5) { u2v t; asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(t) : "v"(lane * 8), "s"(base + 512 * (j & 7)) : "memory"); a2 += t; }
			else if (KIND == 6) { u4v t; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(lane * 16), "s"(base + 1024 * (j & 7)) : "memory"); a4 += t; }
			else if (KIND == 7) { uint32_t t; asm volatile("global_load_dword %0, %1, %2" : "=v"(t) : "v"(lane * 4), "s"(base + 256 * j) : "memory"); a2.x += t; }
			else if (KIND == 8) { uint32_t t; asm volatile("global_load_dword %0, %1, %2" : "=v"(t) : "v"(lane * 8 + (j & 1) * 4), "s"(base + 512 * (j >> 1)) : "memory"); a2.x += t; }	// 8-byte stride: re / im of an fp32 pair separately
			else if (KIND == 9) { u2v t; asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(t) : "v"(lane * 8), "s"(base + 512 * (j & 7)) : "memory"); a2 += t; }
			else if (KIND == 10) { u2v t; asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(t) : "v"(lane * 16), "s"(base + 1024 * (j & 7)) : "memory"); a2 += t; }	// 8 of every 16 bytes
			else if (KIND == 11) { u2v t = { 0, 0 }; if (lane < 32) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a4) : "v"(lane * 16), "s"(base + 512 * (j & 7)) : "memory"); a2 += t; }	// half the lanes, 16 B each: the same 512 B
		}
		if (KIND >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (a2.x + a2.y + a4.x + a4.y + a4.z + a4.w == 0x12345u) sink[0] = 1;
}

int main()
{
	char *d; uint32_t *sink; (void)hipMalloc(&d, 256 * 65536); (void)hipMemset(d, 0, 256 * 65536); (void)hipMalloc(&sink, 64);
	const char *names[] = {"global_store_short, 2 B per 4", "global_store_short dense", "global_store_dword", "global_store_dwordx2", "global_store_dwordx4",
	                       "global_load_dwordx2 (+wait per 16)", "global_load_dwordx4 (+wait per 16)", "global_load_dword (+wait per 16)", "global_load_dword, 8-byte lane stride", "global_load_dwordx2 nt", "global_load_dwordx2, 16-byte lane stride", "global_load_dwordx4, 32 lanes"};
	for (int kind = 0; kind < 12; kind++) {
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		const int iters = 2000;
		auto launch = [&]() {
			switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, d, iters, sink); break;
			C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
			}
		};
		launch(); (void)hipDeviceSynchronize();
		(void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1);
		const double ns = ms * 1e6 / ((double)iters * 16 * 8);		// per instruction and CU
		printf("%-38s %.2f ns per wave-instruction and CU (%.1f cycles at 2.4 GHz)\n", names[kind], ns, ns * 2.4);
	}
	return 0;
}
