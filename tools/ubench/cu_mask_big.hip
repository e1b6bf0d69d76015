// How does a grid of BIG work-groups (512 threads, 135 KiB of LDS, 251 VGPRs: one per CU, like k1w_fft_bin) spread over a stream
// confined by hipExtStreamCreateWithCUMask?  Every work-group spins for 100 us and records its CU and its start time; the host prints the
// kernel's duration, the number of distinct CUs and the most work-groups any CU ran.
// hipcc --offload-arch=gfx950 -O2 -o cu_mask_big cu_mask_big.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>

__global__ __launch_bounds__(512, 2) void big(unsigned long long *out, long long spin)
{
	extern __shared__ unsigned char lds[];
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("v_mov_b32 v250, 0" ::: "v250");
	const long long t0 = wall_clock64();
	lds[threadIdx.x] = (unsigned char)t0;
	while (wall_clock64() - t0 < spin) { }
	if (threadIdx.x == 0) {
		out[2 * blockIdx.x] = ((unsigned long long)(xcc & 0xf) << 16) | ((hw >> 8) & 0xff);
		out[2 * blockIdx.x + 1] = (unsigned long long)t0;
	}
}

static void run(const char *name, hipStream_t st, int blocks)
{
	unsigned long long *d, *h = (unsigned long long *)malloc(blocks * 16);
	hipMalloc(&d, blocks * 16);
	const int lds = 135 * 1024;
	hipFuncSetAttribute(reinterpret_cast<const void *>(big), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int rep = 0; rep < 2; rep++) {
		hipEventRecord(e0, st);
		hipLaunchKernelGGL(big, dim3(blocks), dim3(512), lds, st, d, 10000LL /* 100 us at 100 MHz */);
		hipEventRecord(e1, st);
		hipStreamSynchronize(st);
	}
	float ms; hipEventElapsedTime(&ms, e0, e1);
	hipMemcpy(h, d, blocks * 16, hipMemcpyDeviceToHost);
	std::map<unsigned long long, int> per_cu;
	unsigned long long tmin = ~0ull, tmax = 0;
	for (int i = 0; i < blocks; i++) { per_cu[h[2 * i]]++; if (h[2 * i + 1] < tmin) tmin = h[2 * i + 1]; if (h[2 * i + 1] > tmax) tmax = h[2 * i + 1]; }
	int mx = 0; for (auto &kv : per_cu) if (kv.second > mx) mx = kv.second;
	printf("%-34s grid %3d: %6.1f us, %3zu distinct CUs, at most %d work-groups on one CU, last start %.1f us after the first (err %d)\n",
	       name, blocks, ms * 1e3, per_cu.size(), mx, (double)(tmax - tmin) / 100.0, (int)hipGetLastError());
	hipFree(d); free(h);
}

int main()
{
	hipStream_t s0; hipStreamCreate(&s0);
	run("no mask", s0, 256);
	run("no mask", s0, 248);
	struct { const char *name; uint32_t m[8]; int grid; } masks[] = {
		{ "all 256 bits",              { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u }, 256 },
		{ "bits 0..247",               { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0x00ffffffu }, 248 },
		{ "bits 0..247",               { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0x00ffffffu }, 124 },
		{ "bits 0..239",               { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0x0000ffffu }, 240 },
		{ "bits 0..223",               { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0u }, 224 },
		{ "all but XCD 7 (bit i%8==7)", { 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu }, 224 },
		{ "bits 8..255",               { 0xffffff00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u }, 248 },
	};
	for (auto &mk : masks) {
		hipStream_t st;
		if (hipExtStreamCreateWithCUMask(&st, 8, mk.m) != hipSuccess) { printf("%s: stream creation failed\n", mk.name); continue; }
		run(mk.name, st, mk.grid);
		hipStreamDestroy(st);
	}
	return 0;
}
