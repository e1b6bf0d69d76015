// Microbenchmark: cost of K1's per-sample epilogue (log-power, bin guess, ambiguity, pack, live, max) as the
// compiler schedules it, per spectrum (16 samples per lane), for 1 and 2 waves per SIMD; and with pieces removed.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off epi_rate.hip -o epi_rate && ./epi_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MASK>	// bit0: log, bit1: v/rint/amb, bit2: pack (min+cvt), bit3: live+max
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float seed, float A, float C, float kappa, float w)
{
	v2f x[16];
	float live[16], vmax[16];
	uint32_t pack[16];
	for (int m = 0; m < 16; m++) { x[m] = v2f{seed + threadIdx.x + m, seed * 0.5f + m}; live[m] = 0; vmax[m] = -1e30f; pack[m] = 0; }
	uint32_t amb = 0;
	long long t0 = clock64();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int m = 0; m < 16; m++) {
			const float s = __builtin_fmaf(x[m].x, x[m].x, x[m].y * x[m].y);
			const float l2 = (MASK & 1) ? __builtin_amdgcn_logf(s) : s * 0.001f;
			float r = l2;
			if (MASK & 2) {
				const float v = __builtin_fmaf(A, l2, C);
				r = __builtin_rintf(v);
				const float a = __builtin_fmaf(__builtin_fabsf(l2), kappa, __builtin_fabsf(v - r));
				const uint32_t ab = __float_as_uint(a);
				amb = amb > ab ? amb : ab;
			}
			if (MASK & 4)
				pack[m] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(r, 255.0f), it & 3, pack[m]);
			if (MASK & 8) {
				live[m] = __builtin_fmaf(live[m], w, l2);
				float mx; asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(vmax[m]), "v"(l2)); vmax[m] = mx;
			}
			x[m].x += 1.0f;	// new data each iteration (1 extra VALU per sample)
		}
	}
	long long t1 = clock64();
	float acc = 0; for (int m = 0; m < 16; m++) acc += live[m] + vmax[m] + (float)pack[m] + x[m].y;
	if (acc == 1234.5f || amb == 77) out[0] = acc;
	if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0) / (float)iters;
}

template <int MASK> static void run(float *d, const char *name)
{
	for (int wps = 1; wps <= 2; wps++) {
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		const int iters = 4000;
		hipLaunchKernelGGL(k<MASK>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f, 3.1f, 200.0f, 1e-6f, 0.998f); hipDeviceSynchronize();
		hipEventRecord(e0); hipLaunchKernelGGL(k<MASK>, dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f, 3.1f, 200.0f, 1e-6f, 0.998f); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		float h[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		// per SIMD: wps waves * iters spectra-equivalents
		printf("%-34s waves/SIMD %d: %.0f ns per 16-sample epilogue per SIMD (= %.0f cyc @2.4GHz); wave-local clock %.0f per iteration\n",
		       name, wps, ms * 1e6 / (iters * wps), ms * 1e6 / (iters * wps) * 2.4, h[1]);
	}
}

int main()
{
	float *d; hipMalloc(&d, 64 * sizeof(float)); hipMemset(d, 0, 64 * sizeof(float));
	run<15>(d, "full epilogue");
	run<14>(d, "no v_log (mul instead)");
	run<13>(d, "no v/rint/amb");
	run<11>(d, "no pack (min+cvt_pk_u8)");
	run<7>(d, "no live/max");
	run<1>(d, "s + log only");
	run<0>(d, "s only (mul, fma, mul, add)");
	return 0;
}
