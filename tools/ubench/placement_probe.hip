// Microbenchmark: K1's memory traffic (its 16-byte non-temporal loads one spectrum ahead, its bin dwords every four spectra, its tile
// partials) against WHERE its buffers were allocated, for two ways of dealing the spectra to the waves:
//   tile    K1 today: wave w walks spectra 64 w .. 64 w + 63 -- 1024 read streams 512 KiB apart, 1024 write streams 64 KiB apart
//   quads   wave w takes the quads of spectra w, w + 1024, w + 2048, ... -- at any moment the launch touches ONE window of 32 MiB
//           of IQ and 4 MiB of bin dwords (what a K1 with interleaved tiles would do; needs other tile-partial weights)
// A fresh (IQ, bins, partials) triple is allocated per round and kept, so that the allocator hands out different memory each time:
// on a box with both kinds of memory the `tile` column shows two values (98 / 109.5 us), and the question is whether `quads` does.
// hipcc --offload-arch=gfx950 -O3 placement_probe.hip -o placement_probe && ./placement_probe [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int QUADS>
__global__ __launch_bounds__(256, 2) void k(const v4f *__restrict__ src, float *out, unsigned *bins, float2 *partial)
{
	const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
	auto spec = [&](int k) -> size_t {			/* k-th spectrum this wave processes */
		if (QUADS) return ((size_t)(k >> 2) * 1024 + w) * 4 + (k & 3);
		return (size_t)w * 64 + k;
	};
	v4f cur[8], nxt[8];
	v4f acc = {0, 0, 0, 0};
#pragma unroll
	for (int j = 0; j < 8; j++) nxt[j] = __builtin_nontemporal_load(src + spec(0) * 512 + lane + 64 * j);
	for (int kk = 0; kk < 64; kk++) {
#pragma unroll
		for (int j = 0; j < 8; j++) cur[j] = nxt[j];
		if (kk + 1 < 64) {
#pragma unroll
			for (int j = 0; j < 8; j++) nxt[j] = __builtin_nontemporal_load(src + spec(kk + 1) * 512 + lane + 64 * j);
		}
#pragma unroll
		for (int j = 0; j < 8; j++) acc += cur[j];
		if ((kk & 3) == 3) {			/* the quad's bin dwords: 4 KiB, row = quad index */
			unsigned *dst = bins + (spec(kk) >> 2) * 1024 + lane;
#pragma unroll
			for (int m = 0; m < 16; m++) dst[64 * m] = __float_as_uint(acc.x) + m;
		}
	}
	float2 *pp = partial + (size_t)w * 1024 + lane;
#pragma unroll
	for (int m = 0; m < 16; m++) pp[64 * m] = make_float2(acc.y, acc.z + m);
	if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x;
}

int main(int argc, char **argv)
{
	const int rounds = argc > 1 ? atoi(argv[1]) : 8;
	const size_t per = (size_t)65536 * 8192;		// one launch: 65536 spectra = 512 MiB of IQ
	float *out; hipMalloc(&out, 64);
	hipStream_t st[2]; hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
	for (int r = 0; r < rounds; r++) {
		v4f *src; unsigned *bins; float2 *partial;
		if (hipMalloc(&src, per) != hipSuccess || hipMalloc(&bins, (size_t)64 << 20) != hipSuccess || hipMalloc(&partial, (size_t)8 << 20) != hipSuccess) break;
		hipMemset(src, 0, per);
		float us[2];
		for (int q = 0; q < 2; q++) {
			auto launch = [&](int i) {
				if (q) hipLaunchKernelGGL((k<1>), dim3(256), dim3(256), 0, st[i & 1], src, out, bins, partial);
				else   hipLaunchKernelGGL((k<0>), dim3(256), dim3(256), 0, st[i & 1], src, out, bins, partial);
			};
			for (int i = 0; i < 8; i++) launch(i);
			hipDeviceSynchronize();
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			hipEventRecord(e0, st[0]);
			const int n = 32;
			for (int i = 0; i < n; i++) launch(i);
			hipStreamSynchronize(st[1]);
			hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			us[q] = ms * 1e3f / n;
		}
		printf("allocation %d (IQ at %p): tile %.1f us, quads %.1f us per 512 MiB launch (two in flight)\n", r, (void *)src, us[0], us[1]);
	}
	return 0;
}
