/*
 * fosphor_kernels.hip -- CDNA4 (gfx950) kernels of the fosphor compute core
 *
 * Replaces lib/fosphor/fft.cl + lib/fosphor/display.cl of the reference.
 * Compile with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
 * -ffp-contract=off is REQUIRED: the FFT must round exactly where the reference's
 * expressions round (mul, then add), or histogram counts stop being bit-exact.
 *
 * K1  k1_fft_bin   one WAVE per spectrum (64 lanes x 16 points), no block barriers.
 *                  Radix 8.8.8.2 Stockham with the reference's exact butterfly and
 *                  twiddle order (fft.cl:86-145,278-350,397-466); three exchanges
 *                  through an XOR-swizzled 8 KiB LDS slab per wave; twiddles and
 *                  window live in registers for the whole tile of spectra.
 *                  Epilogue per sample: |X|^2 -> v_log_f32 -> bin guess, accepted
 *                  when provably on the right side of a bin edge, otherwise decided
 *                  by comparing the double-precision |X|^2 with host-computed exact
 *                  thresholds (fosphor_portable_math.h) -- so integer bins equal the
 *                  oracle's log10(hypot()) pipeline bit for bit without evaluating it.
 * K2  k2_count     LDS-privatised histogram per (16-column slab, batch): ds_add on
 *                  [bin][col] (display.cl:161-177), plus the per-batch live sum / max.
 * K3  k3_merge     per (bin, x) cell rise/decay over all batches of the launch in
 *                  order (display.cl:217-254); live EMA + max-hold (display.cl:186-214,
 *                  257-310).
 */
#include "fosphor_internal.h"

#pragma clang fp contract(off)

namespace fosphor_amd {

/* ------------------------------------------------------------------------ */
/* Complex helpers: same operations, same order as fft.cl                   */
/* ------------------------------------------------------------------------ */

#define F_SQRT_1_2 (0.707106781188f)	/* fft.cl:72 */

/* fft.cl:37-46 */
static __device__ __forceinline__ float2 c_mul(float2 a, float2 b)
{
	float2 r;
	r.x = a.x * b.x - a.y * b.y;
	r.y = a.x * b.y + a.y * b.x;
	return r;
}

/* fft.cl:77-82 */
static __device__ __forceinline__ float2 mul_p1q2(float2 a) { return make_float2(a.y, -a.x); }
static __device__ __forceinline__ float2 mul_p1q4(float2 a)
{
	return make_float2(F_SQRT_1_2 * (a.x + a.y), F_SQRT_1_2 * (-a.x + a.y));
}
static __device__ __forceinline__ float2 mul_p3q4(float2 a)
{
	return make_float2(F_SQRT_1_2 * (-a.x + a.y), F_SQRT_1_2 * (-a.x - a.y));
}

/* fft.cl:86-94 */
#define DFT2(a, b) do { \
		float2 _t = make_float2((a).x - (b).x, (a).y - (b).y); \
		(a) = make_float2((a).x + (b).x, (a).y + (b).y); \
		(b) = _t; \
	} while (0)

/* fft.cl:112-145 */
static __device__ __forceinline__ void dft8(float2 (&r)[8])
{
	DFT2(r[0], r[4]); DFT2(r[1], r[5]); DFT2(r[2], r[6]); DFT2(r[3], r[7]);
	r[5] = mul_p1q4(r[5]); r[6] = mul_p1q2(r[6]); r[7] = mul_p3q4(r[7]);
	DFT2(r[0], r[2]); DFT2(r[1], r[3]); DFT2(r[4], r[6]); DFT2(r[5], r[7]);
	r[3] = mul_p1q2(r[3]); r[7] = mul_p1q2(r[7]);
	DFT2(r[0], r[1]); DFT2(r[2], r[3]); DFT2(r[4], r[5]); DFT2(r[6], r[7]);
}

/* Order in which a radix-8 pass stores its outputs: offsets {0,p,..,7p} receive
 * r[0,4,2,6,1,5,3,7] (fft.cl:321-328). */
#define R8_PERM(jj) (((jj) == 0) ? 0 : ((jj) == 1) ? 4 : ((jj) == 2) ? 2 : ((jj) == 3) ? 6 : \
                     ((jj) == 4) ? 1 : ((jj) == 5) ? 5 : ((jj) == 6) ? 3 : 7)

/* Intra-wave LDS exchange: program order within the wave is the only ordering needed */
static __device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ------------------------------------------------------------------------ */
/* Exact binning                                                            */
/* ------------------------------------------------------------------------ */

#define F_HALF_LOG10_2 (0.150514997831990597606869447362f)	/* pwr = log10|X| = this * log2(|X|^2) */

/* Decide a sample the fast path could not: compare |X|^2, formed in double with one
 * rounding (both squares are exact), against the exact thresholds.
 * thr[b] for b in [1, nb) = smallest double s with oracle_bin(s) >= b; thr[0] = -1;
 * thr[nb] = smallest s whose hypot overflows float (-> non-finite -> bin 0,
 * fosphor_portable_math.h fpm_bin_from_pwr). */
static __device__ __forceinline__ uint32_t bin_exact(float re, float im, float pwr_fast, int guess,
                                                      const double *__restrict__ thr, int nb, float *pwr_out)
{
	const double xr = (double)re, xi = (double)im;
	const double sd = __builtin_fma(xr, xr, xi * xi);
	const float  sf = (float)sd;
	int bin;

	if (sf >= 1e-30f && sf <= 1e30f) {
		/* the guess is within one bin of the truth */
		bin = guess - (sd < thr[guess] ? 1 : 0) + (sd >= thr[guess + 1] ? 1 : 0);
		*pwr_out = pwr_fast;
	} else {
		/* zero, denormal, huge, inf or NaN: full search, and a log-power that does not
		 * depend on |X|^2 fitting a float: split sd = m * 2^e, m in [1,2) */
		int lo = 0, hi = nb;		/* invariant: sd >= thr[lo] (thr[0] = -1); sd < thr[hi] or hi == nb */
		if (sd >= thr[nb]) {
			bin = nb;
		} else if (!(sd >= 0.0)) {
			bin = 0;		/* NaN */
		} else {
			while (hi - lo > 1) {
				int mid = (lo + hi) >> 1;
				if (sd >= thr[mid]) lo = mid; else hi = mid;
			}
			bin = lo;
		}
		if (__builtin_isinf(re) || __builtin_isinf(im) || sd >= thr[nb]) {
			*pwr_out = __builtin_inff();		/* hypot(inf, anything) = inf; float hypot overflow */
		} else if (sd == 0.0) {
			*pwr_out = -__builtin_inff();		/* log10(0) */
		} else if (sd != sd) {
			*pwr_out = __builtin_nanf("");
		} else {
			const unsigned long long u = (unsigned long long)__double_as_longlong(sd);
			const int e = (int)((u >> 52) & 0x7ff) - 1023;
			const double m = __longlong_as_double((long long)((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
			*pwr_out = ((float)e + __builtin_amdgcn_logf((float)m)) * F_HALF_LOG10_2;
		}
	}
	if (bin >= nb)
		bin = 0;
	return (uint32_t)bin;
}

struct BinConst { float A, C, amb; int nb; const double *thr; };

/* Fast path.  Returns the bin guess; *ok says whether it is provably exact. */
static __device__ __forceinline__ int bin_fast(float re, float im, const BinConst &k, float *pwr, bool *ok)
{
	const float s  = __builtin_fmaf(re, re, im * im);
	const float l2 = __builtin_amdgcn_logf(s);		/* v_log_f32 */
	const float v  = __builtin_fmaf(k.A, l2, k.C);
	const float r  = __builtin_rintf(v);
	/* clamp in float first (med3; NaN -> 0), so the conversion is always defined */
	const int g = (int)__builtin_amdgcn_fmed3f(r, 0.0f, (float)(k.nb - 1));
	*pwr = l2 * F_HALF_LOG10_2;
	/* confident: well inside a bin, and |X|^2 within [2^-32, 2^32) where the v_log_f32 error
	 * bound used to size `amb` holds -- one unsigned compare on the exponent field:
	 * bits(2^-32) = 0x2f800000, bits(2^32) = 0x4f800000.  NaN / inf / 0 / negative fail it. */
	const unsigned su = __float_as_uint(s);
	*ok = (__builtin_fabsf(v - r) <= k.amb) & ((su - 0x2f800000u) < 0x20000000u);
	return g;
}

/* ------------------------------------------------------------------------ */
/* K1                                                                       */
/* ------------------------------------------------------------------------ */

/* Tunables (tools/ab_bench.sh builds variants with -D...) */
#ifndef K1_WAVES_PER_SIMD
#define K1_WAVES_PER_SIMD 2		/* __launch_bounds__ second argument */
#endif
#ifndef K1_PREFETCH
#define K1_PREFETCH 1			/* register prefetch of the next spectrum */
#endif
#ifndef K1_TW3_LDS
#define K1_TW3_LDS 0			/* pass-3 twiddles from an LDS table instead of registers */
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

/* 16 x (64 lanes x 8 B) coalesced, read-once: non-temporal */
static __device__ __forceinline__ void load_iq16(float2 (&x)[16], const float2 *__restrict__ src)
{
#pragma unroll
	for (int m = 0; m < 16; m++) {
		const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(src + 64 * m));
		x[m] = make_float2(v.x, v.y);
	}
}

template <bool WRITE_FFT>
__global__ __launch_bounds__(256, K1_WAVES_PER_SIMD)
void k1_fft_bin(const K1Params p)
{
	__shared__ float2 lds[4][kN];			/* 8 KiB exchange slab per wave */
	__shared__ float2 tw4_tab[512];			/* pass-4 twiddles, shared by the block */
	__shared__ float  win_tab[kN];			/* window, shared by the block */
#if K1_TW3_LDS
	__shared__ float2 tw3_tab[7][64];		/* pass-3 twiddles [n-1][k] */
#endif

	const int lane   = threadIdx.x & 63;
	const int wv     = threadIdx.x >> 6;
	const int ntiles = p.total / p.tile;
	const int stride = gridDim.x * 4;		/* waves in the grid */
	int tile = blockIdx.x * 4 + wv;

	for (int i = threadIdx.x; i < kN; i += 256)
		win_tab[i] = p.win[i];
	for (int i = threadIdx.x; i < 512; i += 256)
		tw4_tab[i] = p.tw[kTw4Off + i];
#if K1_TW3_LDS
	for (int i = threadIdx.x; i < 7 * 64; i += 256)
		tw3_tab[i % 7][i / 7] = p.tw[kTw3Off + i];
#endif
	__syncthreads();				/* the only block-wide barrier */

	if (tile >= ntiles)
		return;					/* whole wave leaves */

	float2 *buf = lds[wv];

	/* ---- per-lane constants, loaded once per tile ------------------------- */
	float2 tw2[7];
#if !K1_TW3_LDS
	float2 tw3[7];
#endif
#pragma unroll
	for (int n = 0; n < 7; n++) {
		tw2[n] = p.tw[kTw2Off + (lane & 7) * 7 + n];	/* k = i & 7  (both virtual items) */
#if !K1_TW3_LDS
		tw3[n] = p.tw[kTw3Off + lane * 7 + n];		/* k = i & 63 = lane               */
#endif
	}

	/* ---- swizzled LDS addressing -------------------------------------------
	 * element e lives at phys(e) = e ^ ((e >> 3) & 15): every access below is
	 * bank-conflict free for ds_read_b64 (32-lane groups, 64 banks) and
	 * ds_write_b64 (16-lane groups, 32 banks).  The closed forms per access
	 * pattern are derived in DESIGN.md ("LDS exchange").                    */
	const int rd_even = lane ^ ((lane >> 3) & 7);		/* e = lane + 64m, m even */
	const int rd_odd  = rd_even ^ 8;			/*                 m odd  */
	const int st1     = (8 * lane) ^ (lane & 15);		/* pass 1: e = 8i + jj    */
	const int st2     = ((64 * (lane >> 3)) + (lane & 7)) ^ (lane & 8);	/* pass 2: e = 64(i>>3)+(i&7)+8jj */

	const BinConst bk = { p.binA, p.binC, p.amb, p.n_bins, p.thr };

	float2 xn[16];
#if K1_PREFETCH
	load_iq16(xn, p.iq + (size_t)tile * p.tile * kN + lane);
#endif

	/* persistent wave: tiles tile, tile + stride, ... (per-lane constants stay in registers) */
	for (; tile < ntiles; tile += stride) {
	const int t0 = tile * p.tile;

	float live[16], vmax[16];
#pragma unroll
	for (int m = 0; m < 16; m++) {
		live[m] = 0.0f;
		vmax[m] = -1000.0f;				/* display.cl:91 */
	}

	for (int g0 = 0; g0 < p.tile; g0 += 4) {
		uint32_t pack[16];
#pragma unroll
		for (int m = 0; m < 16; m++)
			pack[m] = 0;

#pragma unroll 1
		for (int u = 0; u < 4; u++) {
			const int t = t0 + g0 + u;
			float2 x[16];

#if !K1_PREFETCH
			load_iq16(xn, p.iq + (size_t)t * kN + lane);
#endif
			/* window (fft.cl:415-417) */
#pragma unroll
			for (int m = 0; m < 16; m++)
			{
				const float w = win_tab[lane + 64 * m];
				x[m] = make_float2(xn[m].x * w, xn[m].y * w);
			}

#if K1_PREFETCH
			/* prefetch the next spectrum this wave will process */
			{
				const bool last = (g0 + u + 1 == p.tile);
				const int t_next = last ? (tile + stride) * p.tile : t + 1;
				if (!last || tile + stride < ntiles)
					load_iq16(xn, p.iq + (size_t)t_next * kN + lane);
			}
#endif

			/* ---- pass 1: radix 8, p = 1, no twiddle (fft.cl:419-420) --------
			 * virtual work-item i = lane + 64v owns elements i + 128j = lane + 64(v + 2j) */
#pragma unroll
			for (int v = 0; v < 2; v++) {
				float2 r[8];
#pragma unroll
				for (int j = 0; j < 8; j++)
					r[j] = x[v + 2 * j];
				dft8(r);
#pragma unroll
				for (int jj = 0; jj < 8; jj++)
					buf[(st1 ^ jj) + 512 * v] = r[R8_PERM(jj)];
			}
			wave_lds_sync();
#pragma unroll
			for (int m = 0; m < 16; m++)
				x[m] = buf[((m & 1) ? rd_odd : rd_even) + 64 * m];
			wave_lds_sync();

			/* ---- pass 2: radix 8, p = 8 (fft.cl:422-423) ------------------- */
#pragma unroll
			for (int v = 0; v < 2; v++) {
				float2 r[8];
				r[0] = x[v];
#pragma unroll
				for (int j = 1; j < 8; j++)
					r[j] = c_mul(x[v + 2 * j], tw2[j - 1]);
				dft8(r);
#pragma unroll
				for (int jj = 0; jj < 8; jj++)
					buf[(st2 ^ (9 * jj)) + 512 * v] = r[R8_PERM(jj)];
			}
			wave_lds_sync();
#pragma unroll
			for (int m = 0; m < 16; m++)
				x[m] = buf[((m & 1) ? rd_odd : rd_even) + 64 * m];
			wave_lds_sync();

			/* ---- pass 3: radix 8, p = 64 (fft.cl:425-426) ------------------ */
#pragma unroll
			for (int v = 0; v < 2; v++) {
				float2 r[8];
				r[0] = x[v];
#pragma unroll
				for (int j = 1; j < 8; j++)
#if K1_TW3_LDS
					r[j] = c_mul(x[v + 2 * j], tw3_tab[j - 1][lane]);
#else
					r[j] = c_mul(x[v + 2 * j], tw3[j - 1]);
#endif
				dft8(r);
#pragma unroll
				for (int jj = 0; jj < 8; jj++)	/* e = 512v + lane + 64jj */
					buf[((jj & 1) ? rd_odd : rd_even) + 64 * jj + 512 * v] = r[R8_PERM(jj)];
			}
			wave_lds_sync();
#pragma unroll
			for (int m = 0; m < 16; m++)
				x[m] = buf[((m & 1) ? rd_odd : rd_even) + 64 * m];
			wave_lds_sync();

			/* ---- pass 4: radix 2, p = 512 (fft.cl:428-458) ------------------
			 * butterfly on elements (j, j + 512), j = lane + 64c, twiddle k = j.
			 * Results: X[j] -> x[c], X[j + 512] -> x[c + 8], i.e. column lane + 64m. */
#pragma unroll
			for (int c = 0; c < 8; c++) {
				float2 a = x[c];
				float2 b = c_mul(x[c + 8], tw4_tab[lane + 64 * c]);	/* k = lane + 64c */
				DFT2(a, b);
				x[c] = a;
				x[c + 8] = b;
			}

			if (WRITE_FFT) {
#pragma unroll
				for (int m = 0; m < 16; m++)
					p.fft_out[(size_t)t * kN + lane + 64 * m] = x[m];
			}

			/* ---- epilogue: log-power, exact bin (display.cl:136,161-168) ----
			 * in chunks of 4 columns to keep the live register set small */
			const bool store_row = (t >= p.wf_first);
			float *wf_row = p.wf + (size_t)((p.wf_pos0 + t) & p.wf_mask) * kN + lane;
#pragma unroll
			for (int m0 = 0; m0 < 16; m0 += 4) {
				float    pw[4];
				uint32_t bn[4];
				uint32_t redo = 0;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					bool ok;
					bn[q] = (uint32_t)bin_fast(x[m0 + q].x, x[m0 + q].y, bk, &pw[q], &ok);
					redo |= ok ? 0u : (1u << q);
				}
				while (redo) {		/* rare: ~2e-4 of samples; one code copy per chunk */
					const int q = __builtin_ctz(redo);
					const float    sre = q == 0 ? x[m0].x : q == 1 ? x[m0 + 1].x : q == 2 ? x[m0 + 2].x : x[m0 + 3].x;
					const float    sim = q == 0 ? x[m0].y : q == 1 ? x[m0 + 1].y : q == 2 ? x[m0 + 2].y : x[m0 + 3].y;
					const float    spw = q == 0 ? pw[0] : q == 1 ? pw[1] : q == 2 ? pw[2] : pw[3];
					const uint32_t sbn = q == 0 ? bn[0] : q == 1 ? bn[1] : q == 2 ? bn[2] : bn[3];
					float npw;
					const uint32_t nbn = bin_exact(sre, sim, spw, (int)sbn, bk.thr, bk.nb, &npw);
#pragma unroll
					for (int qq = 0; qq < 4; qq++) {
						bn[qq] = (q == qq) ? nbn : bn[qq];
						pw[qq] = (q == qq) ? npw : pw[qq];
					}
					redo &= redo - 1;
				}
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const int m = m0 + q;
					pack[m] |= bn[q] << (8 * u);
					live[m] = __builtin_fmaf(live[m], p.w, pw[q]);		/* Horner form of display.cl:149-150 (tolerance-checked float) */
					vmax[m] = __builtin_fmaxf(vmax[m], pw[q]);		/* = OpenCL max() here: NaN pwr is ignored, display.cl:139 */
					if (store_row)
						wf_row[64 * m] = pw[q];				/* display.cl:142-146 */
				}
			}
		}

		/* 4 spectra x 1 column per dword, coalesced 256 B per instruction */
		uint32_t *dst = p.bins + (size_t)((t0 + g0) >> 2) * kN + lane;
#pragma unroll
		for (int m = 0; m < 16; m++)
			dst[64 * m] = pack[m];
	}

	float2 *pp = p.partial + (size_t)tile * kN + lane;
#pragma unroll
	for (int m = 0; m < 16; m++)
		pp[64 * m] = make_float2(live[m], vmax[m]);
	}	/* tile loop */
}

hipError_t launch_k1(const K1Params &p, hipStream_t s)
{
	const int tiles  = p.total / p.tile;
	int blocks = (tiles + 3) / 4;
	if (blocks > kK1MaxBlocks)
		blocks = kK1MaxBlocks;		/* persistent: 2 work-groups per CU */
	if (p.fft_out)
		hipLaunchKernelGGL(k1_fft_bin<true>, dim3(blocks), dim3(256), 0, s, p);
	else
		hipLaunchKernelGGL(k1_fft_bin<false>, dim3(blocks), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* Test hook: the K1 epilogue alone on FFT values read from memory */
__global__ __launch_bounds__(256)
void k_bin_hook(const float2 *__restrict__ fft, uint8_t *__restrict__ bin, float *__restrict__ pwr, int n,
                const K1Params p, int force_exact)
{
	const BinConst bk = { p.binA, p.binC, p.amb, p.n_bins, p.thr };
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		float2 v = fft[i];
		float pw; bool ok;
		uint32_t b = (uint32_t)bin_fast(v.x, v.y, bk, &pw, &ok);
		if (!ok || force_exact)
			b = bin_exact(v.x, v.y, pw, (int)b, bk.thr, bk.nb, &pw);
		bin[i] = (uint8_t)b;
		pwr[i] = pw;
	}
}

hipError_t launch_bin_hook(const float2 *fft, uint8_t *bin, float *pwr, int n,
                           const K1Params &p, int force_exact, hipStream_t s)
{
	int blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(k_bin_hook, dim3(blocks), dim3(256), 0, s, fft, bin, pwr, n, p, force_exact);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* K2: hit counts per (slab of 16 columns, batch)                            */
/* ------------------------------------------------------------------------ */

__global__ __launch_bounds__(256)
void k2_count(const K2Params p)
{
	__shared__ uint32_t h[256 * 16];		/* [bin][col], display.cl:96,176 */
	__shared__ float red_s[16][17], red_m[16][17];

	const int tid  = threadIdx.x;
	const int col  = tid & 15;
	const int row  = tid >> 4;
	const int x0   = blockIdx.x * 16;
	const int c    = blockIdx.y;			/* chunk index within the launch */
	const int cpb  = p.batch / p.chunk;		/* chunks per batch */
	const int f    = c / cpb;			/* batch index */
	const int t_in = (c - f * cpb) * p.chunk;	/* first spectrum of the chunk within its batch */
	const int nb   = p.n_bins;

	for (int i = tid; i < nb * 16; i += 256)
		h[i] = 0;
	__syncthreads();

	/* bins: one dword = 4 consecutive spectra of one column */
	const uint32_t *src = p.bins + (size_t)c * (p.chunk >> 2) * kN + x0 + col;
	const int nq = p.chunk >> 2;
	int q = row;
	for (; q + 48 < nq; q += 64) {			/* 4 independent loads in flight per thread */
		uint32_t v[4];
#pragma unroll
		for (int u = 0; u < 4; u++)
			v[u] = src[(size_t)(q + 16 * u) * kN];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			atomicAdd(&h[((v[u]      ) & 0xff) * 16 + col], 1u);
			atomicAdd(&h[((v[u] >>  8) & 0xff) * 16 + col], 1u);
			atomicAdd(&h[((v[u] >> 16) & 0xff) * 16 + col], 1u);
			atomicAdd(&h[((v[u] >> 24)       ) * 16 + col], 1u);
		}
	}
	for (; q < nq; q += 16) {
		const uint32_t v = src[(size_t)q * kN];
		atomicAdd(&h[((v      ) & 0xff) * 16 + col], 1u);
		atomicAdd(&h[((v >>  8) & 0xff) * 16 + col], 1u);
		atomicAdd(&h[((v >> 16) & 0xff) * 16 + col], 1u);
		atomicAdd(&h[((v >> 24)       ) * 16 + col], 1u);
	}

	/* live sum: sum_t pwr_t (1-a)^(B-1-t) from the tile partials, which hold
	 * sum_{t in tile} pwr_t (1-a)^(t_last - t) (display.cl:149-150) */
	{
		const int tiles = p.chunk / p.tile;
		const float2 *pp = p.partial + (size_t)c * tiles * kN + x0 + col;
		float s = 0.0f, m = -1000.0f;
		for (int j = row; j < tiles; j += 16) {
			const float2 v = pp[(size_t)j * kN];
			const int t_last = p.t_offset + t_in + (j + 1) * p.tile - 1;
			/* (1-a)^k as exp2(k log2(1-a)): relative error ~1e-6 where the weight is not negligible */
			s += v.x * __builtin_amdgcn_exp2f(p.log2_w * (float)(p.weight_batch - 1 - t_last));
			m = (m < v.y) ? v.y : m;
		}
		red_s[row][col] = s;
		red_m[row][col] = m;
	}
	__syncthreads();

	if (tid < 16) {
		float s = 0.0f, m = -1000.0f;
		for (int j = 0; j < 16; j++) {
			s += red_s[j][tid];
			m = (m < red_m[j][tid]) ? red_m[j][tid] : m;
		}
		p.chunk_sum[(size_t)c * kN + x0 + tid] = s;
		p.chunk_max[(size_t)c * kN + x0 + tid] = m;
	}

	uint32_t *dst = p.hc + (size_t)f * nb * kN + x0 + col;
	if (cpb == 1) {
		for (int b = row; b < nb; b += 16)
			dst[(size_t)b * kN] = h[b * 16 + col];
	} else {
		for (int b = row; b < nb; b += 16) {
			const uint32_t v = h[b * 16 + col];
			if (v)
				atomicAdd(&dst[(size_t)b * kN], v);
		}
	}
}

hipError_t launch_k2(const K2Params &p, int n_chunks, hipStream_t s)
{
	hipLaunchKernelGGL(k2_count, dim3(kN / 16, n_chunks), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* fixed-order reduction of the chunk partials of each batch */
__global__ __launch_bounds__(256)
void k2b_reduce(const K2bParams p)
{
	const int gid = blockIdx.x * 256 + threadIdx.x;
	if (gid >= p.n_batches * kN)
		return;
	const int f = gid / kN, x = gid - f * kN;
	float s = 0.0f, m = -1000.0f;
	for (int c = 0; c < p.cpb; c++) {
		const size_t i = (size_t)(f * p.cpb + c) * kN + x;
		s += p.chunk_sum[i];
		m = (m < p.chunk_max[i]) ? p.chunk_max[i] : m;
	}
	p.live_sum[gid] = s;
	p.vmax[gid] = m;
}

hipError_t launch_k2b(const K2bParams &p, hipStream_t s)
{
	const int threads = p.n_batches * kN;
	hipLaunchKernelGGL(k2b_reduce, dim3((threads + 255) / 256), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* K3: state update                                                          */
/* ------------------------------------------------------------------------ */

__global__ __launch_bounds__(256)
void k3_merge(const K3Params p)
{
	const int cells = p.n_bins * kN;
	const int gid = blockIdx.x * 256 + threadIdx.x;
	const float fbatch = (float)p.batch;

	if (gid < cells) {
		/* one (bin, x) cell; batches applied in order (display.cl:217-254).
		 * d and e of display.cl:241-245 depend only on the hit count: with a table
		 * rise[hc] = (d, e) (host-computed with the same powf the oracle uses) the update
		 * is a lookup and display.cl:247,250. */
		float hv = p.hist[gid];
		if (p.rise) {
			/* 8 batches of counts in flight per thread: the loop is otherwise one dependent
			 * HBM/L2 round trip per batch */
			int f = 0;
			for (; f + 8 <= p.n_batches; f += 8) {
				uint32_t hc[8];
#pragma unroll
				for (int u = 0; u < 8; u++)
					hc[u] = __builtin_nontemporal_load(&p.hc[(size_t)(f + u) * cells + gid]);
#pragma unroll
				for (int u = 0; u < 8; u++) {
					if (!((hv <= 0.01f) && (hc[u] == 0))) {	/* display.cl:237-238 */
						const float2 de = p.rise[hc[u]];
						hv = (hv - de.x) * de.y + de.x;		/* display.cl:247 */
						hv = (hv < 0.0f) ? 0.0f : hv;		/* clamp, display.cl:250 */
						hv = (1.0f < hv) ? 1.0f : hv;
					}
				}
			}
			for (; f < p.n_batches; f++) {
				const uint32_t hc = p.hc[(size_t)f * cells + gid];
				if (!((hv <= 0.01f) && (hc == 0))) {
					const float2 de = p.rise[hc];
					hv = (hv - de.x) * de.y + de.x;
					hv = (hv < 0.0f) ? 0.0f : hv;
					hv = (1.0f < hv) ? 1.0f : hv;
				}
			}
		} else {
			const float rt0r = 1.0f / p.t0r, rt0d = 1.0f / p.t0d;
			for (int f = 0; f < p.n_batches; f++) {
				const uint32_t hc = p.hc[(size_t)f * cells + gid];
				if ((hv <= 0.01f) && (hc == 0))			/* display.cl:237-238 */
					continue;
				const float a = (float)hc / fbatch;		/* display.cl:241-245 */
				const float b = a * rt0r;
				const float c = b + rt0d;
				const float d = b * (1.0f / c);
				const float e = powf(1.0f - c, fbatch);
				hv = (hv - d) * e + d;				/* display.cl:247 */
				hv = (hv < 0.0f) ? 0.0f : hv;			/* clamp, display.cl:250 */
				hv = (1.0f < hv) ? 1.0f : hv;
			}
		}
		p.hist[gid] = hv;
	} else if (gid < cells + kN) {
		/* one column: live EMA (display.cl:186-214) and max-hold (display.cl:257-310) */
		const int x = gid - cells;
		const int half = kN >> 1;
		const int i = x ^ half;
		const float decay = p.live_decay;
		float live = p.spectrum[i].y;
		float mh   = p.spectrum[kN + i].y;
		for (int f = 0; f < p.n_batches; f++) {
			const float sum = p.live_sum[(size_t)f * kN + x];
			const float mx  = p.vmax[(size_t)f * kN + x];
			if (!__builtin_isfinite(live))
				live = sum / 16.0f;			/* display.cl:206-207 */
			live = live * decay + sum * p.alpha;		/* display.cl:210-211 */
			if (!__builtin_isfinite(mh))
				mh = -3.402823466e+38f;			/* display.cl:290-291 */
			mh = mh * 0.999f + 0.001f * live;		/* display.cl:303 */
			mh = (mh < mx) ? mx : mh;			/* display.cl:304-305 */
		}
		const float vx = ((float)i / (float)half) - 1.0f;	/* display.cl:209,293 */
		p.spectrum[i]      = make_float2(vx, live);
		p.spectrum[kN + i] = make_float2(vx, mh);
	}
}

hipError_t launch_k3(const K3Params &p, hipStream_t s)
{
	const int threads = p.n_bins * kN + kN;
	hipLaunchKernelGGL(k3_merge, dim3((threads + 255) / 256), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */

__global__ void k_fill(float *dst, float value, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = value;
}

hipError_t launch_fill(float *dst, float value, size_t n, hipStream_t s)
{
	size_t blocks = (n + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, dst, value, n);
	return hipGetLastError();
}

} // namespace fosphor_amd
