// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  (MI355X: 8 XCDs x 32 CUs.)
// Each work-group records (XCC_ID, SE_ID, CU_ID) from the hardware registers; the host prints, per mask, how many distinct
// CUs of each XCD were used.  Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <vector>

__global__ void probe(unsigned *out, int spin)
{
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	// keep the work-group resident for a while so that the grid spreads over every CU the queue may use
	long long t0 = wall_clock64();
	while (wall_clock64() - t0 < spin) { }
	if (threadIdx.x == 0)
		out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}

static void run(const char *name, hipStream_t st, unsigned *d, unsigned *h, int blocks)
{
	hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, st, d, 20000 /* 200 us at 100 MHz */);
	hipStreamSynchronize(st);
	hipMemcpy(h, d, blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
	std::set<unsigned> per_xcc[16];
	for (int i = 0; i < blocks; i++) {
		const unsigned xcc = h[i] >> 16, hw = h[i] & 0xffff;
		// HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
		per_xcc[xcc].insert(hw >> 8);
	}
	int total = 0;
	printf("%-28s CUs used per XCD:", name);
	for (int x = 0; x < 8; x++) { printf(" %2zu", per_xcc[x].size()); total += (int)per_xcc[x].size(); }
	printf("  total %d\n", total);
}

int main(int argc, char **argv)
{
	const int blocks = 4096;
	unsigned *d, *h = (unsigned *)malloc(blocks * sizeof(unsigned));
	hipMalloc(&d, blocks * sizeof(unsigned));
	hipStream_t s0;
	hipStreamCreate(&s0);
	run("no mask", s0, d, h, blocks);
	struct { const char *name; uint32_t m[8]; } masks[] = {
		{ "bits 0..223",        { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0 } },
		{ "bits 224..255",      { 0, 0, 0, 0, 0, 0, 0, ~0u } },
		{ "bits 0..31",         { ~0u, 0, 0, 0, 0, 0, 0, 0 } },
		{ "bits 0..7",          { 0xffu, 0, 0, 0, 0, 0, 0, 0 } },
		{ "bit 0 of each byte", { 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u } },
		{ "bits 0..239",        { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0xffffu } },
		{ "bits 240..255",      { 0, 0, 0, 0, 0, 0, 0, 0xffff0000u } },
	};
	for (auto &mk : masks) {
		hipStream_t st;
		hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mk.m);
		if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask: %s\n", mk.name, hipGetErrorString(e)); continue; }
		run(mk.name, st, d, h, blocks);
		hipStreamDestroy(st);
	}
	return 0;
}
