// Microbenchmark: does the ORDER in which a K1 wave walks its tile of 64 consecutive spectra matter to HBM?
// Two launches of 1024 waves (256 work-groups x 4) run concurrently on two streams, like the library's sub-launches; every wave
// reads 64 spectra of 8 KiB with 16-byte non-temporal loads, the next spectrum requested before the current one is consumed.
//   pattern 0  K1 today: wave w reads spectra 64 w + k, k = 0..63 (all waves in phase: addresses 512 KiB apart)
//   pattern 1  the same tile, started at a wave-dependent phase: k' = (k + rot(w)) & 63, rot = 4 ((13 w) & 15)
//   pattern 2  streaming: wave w reads spectra 1024 k + w (not usable by K1: a wave must own consecutive spectra)
//   pattern 3  phase by work-group: rot = 4 ((5 blockIdx) & 15)
// hipcc --offload-arch=gfx950 -O3 read_skew.hip -o read_skew && ./read_skew
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int PATTERN, int WRITES = 0>
__global__ __launch_bounds__(256, 2) void k(const v4f *__restrict__ src, float *out, unsigned *bins = nullptr, float2 *partial = nullptr)
{
	const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
	int rot = 0;
	if (PATTERN == 1) rot = 4 * ((13 * w) & 15);
	if (PATTERN == 3) rot = 4 * ((5 * (int)blockIdx.x) & 15);
	auto spec = [&](int k) -> size_t {
		if (PATTERN == 2) return (size_t)k * 1024 + w;
		return (size_t)w * 64 + ((k + rot) & 63);
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
		if (WRITES && (kk & 3) == 3) {		/* K1's bin dwords: 4 spectra x 1024 columns = 4 KiB per quad */
			unsigned *dst = bins + ((size_t)w * 16 + (kk >> 2)) * 1024 + lane;
#pragma unroll
			for (int m = 0; m < 16; m++) dst[64 * m] = __float_as_uint(acc.x) + m;
		}
	}
	if (WRITES) {					/* K1's tile partials: 8 KiB per tile */
		float2 *pp = partial + (size_t)w * 1024 + lane;
#pragma unroll
		for (int m = 0; m < 16; m++) pp[64 * m] = make_float2(acc.y, acc.z + m);
	}
	if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x;
}

int main()
{
	const size_t per = (size_t)65536 * 8192;		// one launch: 65536 spectra = 512 MiB
	v4f *src; float *out;
	hipMalloc(&src, 4 * per); hipMalloc(&out, 64);		// 2 GiB ring: 4 launches' worth
	hipMemset(src, 0, 4 * per);
	hipStream_t st[2]; hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
	unsigned *bins[3]; float2 *partial[3];
	for (int i = 0; i < 3; i++) { hipMalloc(&bins[i], (size_t)64 << 20); hipMalloc(&partial[i], (size_t)8 << 20); }
	const char *names[4] = { "tile, in phase (K1 today)", "tile, phase per wave", "streaming (not usable)", "tile, phase per work-group" };
	for (int rep = 0; rep < 2; rep++)
	for (int pat = 0; pat < 4; pat++) {
		auto launch = [&](int i) {
			const v4f *s = src + (size_t)(i & 3) * (per / 16);
			hipStream_t q = st[i & 1];
			if (pat == 0)      hipLaunchKernelGGL((k<0>), dim3(256), dim3(256), 0, q, s, out);
			else if (pat == 1) hipLaunchKernelGGL((k<1>), dim3(256), dim3(256), 0, q, s, out);
			else if (pat == 2) hipLaunchKernelGGL((k<2>), dim3(256), dim3(256), 0, q, s, out);
			else               hipLaunchKernelGGL((k<3>), dim3(256), dim3(256), 0, q, s, out);
		};
		for (int i = 0; i < 16; i++) launch(i);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, st[0]);
		const int n = 64;
		for (int i = 0; i < n; i++) launch(i);
		hipStreamSynchronize(st[1]);
		hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("%-30s: %.1f us per 512 MiB launch (two in flight) = %.2f TB/s\n", names[pat], ms * 1e3 / n, 536.870912e6 / (ms * 1e-3 / n) / 1e12);
	}
	/* K1's own traffic: the tile walk + 64 MiB of bin dwords + 8 MiB of partials per launch, three rotating output sets */
	for (int rep = 0; rep < 2; rep++)
	for (int streams = 1; streams <= 2; streams++) {
		auto launch = [&](int i) {
			const v4f *s = src + (size_t)(i & 3) * (per / 16);
			hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(256), 0, st[streams == 2 ? (i & 1) : 0], s, out, bins[i % 3], partial[i % 3]);
		};
		for (int i = 0; i < 16; i++) launch(i);
		hipDeviceSynchronize();
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, st[0]);
		const int n = 64;
		for (int i = 0; i < n; i++) launch(i);
		hipStreamSynchronize(st[1]);
		hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("K1's loads AND stores, %d stream%s: %.1f us per launch = %.2f TB/s read, %.2f TB/s total\n", streams, streams == 2 ? "s (two launches in flight)" : " (back to back)",
		       ms * 1e3 / n, 536.870912e6 / (ms * 1e-3 / n) / 1e12, (536.870912e6 + 75.5e6) / (ms * 1e-3 / n) / 1e12);
	}
	return 0;
}
