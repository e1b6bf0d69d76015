// Microbenchmark: the rate of LDS atomics as the count kernel (k2_count) issues them.
// A work-group of NW waves hammers a [bins][W] counter image in LDS with ds_add_u32 (no return), bins pseudo-random per lane
// (a hash of a per-lane counter: no memory traffic), lane -> column as below.  Reported: cycles per wave-instruction and CU.
//   layout 0  k2_count today: [bin][32] dwords, lanes c and c + 32 share a dword (low / high half): two lanes per bank AND per address
//   layout 1  [bin][64] dwords, one dword per lane: 64 distinct dwords per instruction
//   layout 2  [bin][32] dwords, but only lanes 0..31 active (what one half costs)
//   layout 3  as 1 with ds_add_rtn (returning atomics)
//   layout 4  [bin][64] with 16-bit halves of two SPECTRA packed: one ds_add_u32 counts bin b for spectrum t (low) ... not possible (different bins) -- skipped
//   layout 5  plain ds_write_b32 in the place of the atomic, layout 1 addresses (the LDS pipe's store rate)
// (addresses are fixed per lane and computed outside the loop: an earlier form hashed them inside it and timed the hash)
// hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_atomic_rate && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LAYOUT, int NW>
__global__ __launch_bounds__(64 * NW) void k(unsigned *out, int iters, int nb)
{
	extern __shared__ unsigned h[];
	const int lane = threadIdx.x & 63;
	const int W = (LAYOUT == 0 || LAYOUT == 2) ? 32 : 64;
	for (int i = threadIdx.x; i < nb * W; i += 64 * NW) h[i] = 0;
	__syncthreads();
	unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	const unsigned col = (W == 32) ? (lane & 31) : lane;
	const unsigned inc = (LAYOUT == 0 && (lane & 32)) ? 0x10000u : 1u;
	unsigned acc = 0;
	/* eight fixed pseudo-random (bin, column) addresses per lane, computed outside the timed loop: the loop is the LDS operations only */
	unsigned *ad[8];
#pragma unroll
	for (int u = 0; u < 8; u++) {
		s = s * 1664525u + 1013904223u;
		ad[u] = &h[((s >> 16) & (unsigned)(nb - 1)) * W + col];
	}
	const long long t0 = clock64();
	if (LAYOUT != 2 || lane < 32) {
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int u = 0; u < 8; u++) {
				if (LAYOUT == 3) acc += atomicAdd(ad[u], inc);
				else if (LAYOUT == 5) *(volatile unsigned *)ad[u] = inc;
				else atomicAdd(ad[u], inc);
			}
		}
	}
	__syncthreads();
	const long long t1 = clock64();
	unsigned sum = acc;
	for (int i = threadIdx.x; i < nb * W; i += 64 * NW) sum += h[i];
	if (sum == 0xdeadbeefu) out[0] = sum;
	if (threadIdx.x == 0) reinterpret_cast<long long *>(out + 16)[blockIdx.x] = t1 - t0;
}

template <int LAYOUT, int NW>
static void run(const char *name, int nb)
{
	unsigned *out; hipMalloc(&out, 1 << 20);
	const int W = (LAYOUT == 0 || LAYOUT == 2) ? 32 : 64;
	const size_t lds = (size_t)nb * W * 4;
	hipFuncSetAttribute(reinterpret_cast<const void *>(&k<LAYOUT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	const int iters = 512;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int rep = 0; rep < 3; rep++) {
		hipEventRecord(e0);
		hipLaunchKernelGGL((k<LAYOUT, NW>), dim3(256), dim3(64 * NW), lds, 0, out, iters, nb);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		long long cyc[4]; hipMemcpy(cyc, out + 16, sizeof(cyc), hipMemcpyDeviceToHost);
		if (rep == 2)
			printf("%-44s nb %3d NW %2d LDS %6zu B: %.1f us, %.1f clock64 ticks per wave-instruction and CU (err %d)\n", name, nb, NW, lds, ms * 1e3,
			       (double)cyc[0] / ((double)iters * 8 * NW), (int)hipGetLastError());
	}
	hipFree(out);
}

int main()
{
	run<0, 16>("[bin][32] halves, 64 lanes (today)", 512);
	run<1, 16>("[bin][64] dwords, 64 lanes", 512);
	run<2, 16>("[bin][32], 32 lanes only", 512);
	run<3, 16>("[bin][64] returning atomics", 512);
	run<5, 16>("[bin][64] plain stores", 512);
	run<0, 8>("[bin][32] halves, 64 lanes (today)", 512);
	run<1, 8>("[bin][64] dwords, 64 lanes", 512);
	run<0, 4>("[bin][32] halves (today)", 256);
	run<1, 4>("[bin][64] dwords", 256);
	run<0, 16>("[bin][32] halves, narrow bins", 16);
	run<1, 16>("[bin][64] dwords, narrow bins", 16);
	return 0;
}
