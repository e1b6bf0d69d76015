// Microbenchmark: HBM read rate of K1's access pattern vs a fully streaming one.
//   pattern 0 (K1): wave w owns tiles w, w + W of 16 consecutive 8-KiB spectra; at any moment the W waves
//                   read 8-KiB pieces 128 KiB apart.
//   pattern 1     : at step k the W waves read W consecutive 8-KiB pieces (one contiguous 16-MiB region).
// Both: 16-byte non-temporal loads, 8 per lane per spectrum, next spectrum issued before the current one is consumed.
// hipcc --offload-arch=gfx950 -O3 read_pattern.hip -o read_pattern && ./read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

// WRITES: also emit K1's output traffic: 4 KiB of "bin bytes" per 4 spectra and 8 KiB of "partials" per 16 (1.5 B per sample)
template <int PATTERN, int WRITES>
__global__ __launch_bounds__(256, 2) void k(const v4f *__restrict__ src, float *out, int spectra_per_wave, int total_spectra,
                                            unsigned *bins, float2 *partial)
{
	const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6), W = gridDim.x * 4;
	auto spec = [&](int k) -> size_t {
		if (PATTERN == 0) return (size_t)((w + (k >> 4) * W) * 16 + (k & 15));
		return (size_t)k * W + w;
	};
	v4f cur[8], nxt[8];
	v4f acc = {0, 0, 0, 0};
#pragma unroll
	for (int j = 0; j < 8; j++)
		nxt[j] = __builtin_nontemporal_load(src + spec(0) * 512 + lane + 64 * j);
	for (int k = 0; k < spectra_per_wave; k++) {
#pragma unroll
		for (int j = 0; j < 8; j++) cur[j] = nxt[j];
		if (k + 1 < spectra_per_wave) {
#pragma unroll
			for (int j = 0; j < 8; j++)
				nxt[j] = __builtin_nontemporal_load(src + spec(k + 1) * 512 + lane + 64 * j);
		}
#pragma unroll
		for (int j = 0; j < 8; j++) acc += cur[j];
		if (WRITES) {
			const size_t t = (PATTERN == 0) ? spec(k) : (size_t)w * spectra_per_wave + k;	// consecutive per wave
			if ((k & 3) == 3) {
				unsigned *dst = bins + (t >> 2) * 1024 + lane;
#pragma unroll
				for (int m = 0; m < 16; m++) dst[64 * m] = __float_as_uint(acc.x) + m;
			}
			if ((k & 15) == 15) {
				float2 *pp = partial + (t >> 4) * 1024 + lane;
#pragma unroll
				for (int m = 0; m < 16; m++) pp[64 * m] = make_float2(acc.y, acc.z + m);
			}
		}
	}
	if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x;
}

int main()
{
	const int W = 2048, spw = 32, total = W * spw;		// 65536 spectra of 8 KiB = 512 MiB
	v4f *src; float *out; unsigned *bins; float2 *partial;
	hipMalloc(&src, (size_t)total * 8192); hipMalloc(&out, 64);
	hipMalloc(&bins, (size_t)total / 4 * 4096); hipMalloc(&partial, (size_t)total / 16 * 8192);
	hipMemset(src, 0, (size_t)total * 8192);
	for (int pat = 0; pat < 4; pat++) {
		for (int rep = 0; rep < 2; rep++) {
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			auto launch = [&]() {
				if (pat == 0)      hipLaunchKernelGGL((k<0, 0>), dim3(W / 4), dim3(256), 0, 0, src, out, spw, total, bins, partial);
				else if (pat == 1) hipLaunchKernelGGL((k<1, 0>), dim3(W / 4), dim3(256), 0, 0, src, out, spw, total, bins, partial);
				else if (pat == 2) hipLaunchKernelGGL((k<0, 1>), dim3(W / 4), dim3(256), 0, 0, src, out, spw, total, bins, partial);
				else               hipLaunchKernelGGL((k<1, 1>), dim3(W / 4), dim3(256), 0, 0, src, out, spw, total, bins, partial);
			};
			for (int i = 0; i < 20; i++) launch();
			hipDeviceSynchronize();
			hipEventRecord(e0);
			for (int i = 0; i < 50; i++) launch();
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			printf("%-26s%s: %.1f us per 512 MiB read  = %.2f TB/s read, %.2f TB/s total\n", (pat & 1) ? "streaming" : "K1 tiles",
			       (pat & 2) ? " + 96 MiB of K1-like writes" : "", ms * 1e3 / 50, 536.870912e6 / (ms * 1e-3 / 50) / 1e12,
			       ((pat & 2) ? 637.5e6 : 536.870912e6) / (ms * 1e-3 / 50) / 1e12);
		}
	}
	return 0;
}
