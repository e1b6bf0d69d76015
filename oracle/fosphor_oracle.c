/*
 * fosphor_oracle.c -- CPU restatement of the fosphor compute hot path
 *
 * TEST INFRASTRUCTURE (see fosphor_oracle.h).  Compile with
 *   gcc -O2 -ffp-contract=off -std=gnu99
 * -ffp-contract=off is REQUIRED: results must not depend on FMA fusion.
 *
 * Every function cites the reference lines it follows.  The arithmetic
 * (operand order, rounding points) mirrors the reference exactly so that
 * this file and the reference kernels (oracle/_ref) agree bit for bit when
 * both use the same built-in binding.
 */
#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fosphor_portable_math.h"
#include "fosphor_oracle.h"

typedef struct { float x, y; } cf;

/* ------------------------------------------------------------------------ */
/* FFT  (fft.cl)                                                            */
/* ------------------------------------------------------------------------ */

#define ORACLE_PI_F	(3.141592653589f)	/* fft.cl:26 */
#define ORACLE_SQRT_1_2	(0.707106781188f)	/* fft.cl:72 */

/* fft.cl:37-46 : cmul_1 */
static inline cf o_cmul(cf a, cf b)
{
	cf r;
	r.x = a.x * b.x - a.y * b.y;
	r.y = a.x * b.y + a.y * b.x;
	return r;
}

/* fft.cl:61-69 : twiddle(a, k, alpha), native_sin/native_cos -> portable */
static inline cf o_twiddle(cf a, int k, float alpha)
{
	cf w;
	w.y = fpm_sinf((float)k * alpha);
	w.x = fpm_cosf((float)k * alpha);
	return o_cmul(a, w);
}

/* fft.cl:77-82 */
static inline cf o_mul_p1q2(cf a) { cf r; r.x = a.y; r.y = -a.x; return r; }
static inline cf o_mul_p1q4(cf a)
{
	cf r;
	r.x = ORACLE_SQRT_1_2 * (a.x + a.y);
	r.y = ORACLE_SQRT_1_2 * (-a.x + a.y);
	return r;
}
static inline cf o_mul_p3q4(cf a)
{
	cf r;
	r.x = ORACLE_SQRT_1_2 * (-a.x + a.y);
	r.y = ORACLE_SQRT_1_2 * (-a.x - a.y);
	return r;
}

/* fft.cl:86-94 */
static inline void o_dft2(cf *a, cf *b)
{
	cf t;
	t.x = a->x - b->x;  t.y = a->y - b->y;
	a->x = a->x + b->x; a->y = a->y + b->y;
	*b = t;
}

/* fft.cl:112-145 */
static inline void o_dft8(cf *r)
{
	o_dft2(&r[0], &r[4]); o_dft2(&r[1], &r[5]); o_dft2(&r[2], &r[6]); o_dft2(&r[3], &r[7]);
	r[5] = o_mul_p1q4(r[5]); r[6] = o_mul_p1q2(r[6]); r[7] = o_mul_p3q4(r[7]);
	o_dft2(&r[0], &r[2]); o_dft2(&r[1], &r[3]); o_dft2(&r[4], &r[6]); o_dft2(&r[5], &r[7]);
	r[3] = o_mul_p1q2(r[3]); r[7] = o_mul_p1q2(r[7]);
	o_dft2(&r[0], &r[1]); o_dft2(&r[2], &r[3]); o_dft2(&r[4], &r[5]); o_dft2(&r[6], &r[7]);
}

/* fft.cl:97-110 (dormant in the reference; used for N with log2(N)%3 == 2) */
static inline void o_dft4(cf *r)
{
	o_dft2(&r[0], &r[2]); o_dft2(&r[1], &r[3]);
	r[3] = o_mul_p1q2(r[3]);
	o_dft2(&r[0], &r[1]); o_dft2(&r[2], &r[3]);
}

/* One Stockham radix-8 pass over a whole spectrum: fft.cl:278-350.
 * src and dst are distinct: the reference reads all inputs, barriers, then
 * writes (fft.cl:338-349), which is exactly a ping-pong. */
static void o_pass_radix8(const cf *src, cf *dst, int n, int p, int tw)
{
	const int t = n >> 3;		/* work-group size = N/8 (fft.cl:403) */
	static const int perm[8] = { 0, 4, 2, 6, 1, 5, 3, 7 };	/* fft.cl:321-328 */
	int i, j;

	for (i = 0; i < t; i++) {
		cf r[8];
		int k = i & (p - 1);
		int j0;

		for (j = 0; j < 8; j++)			/* fft.cl:299-312 */
			r[j] = src[i + j * t];

		if (tw) {				/* fft.cl:285-297 */
			float alpha = -ORACLE_PI_F * (float)k / (float)(4 * p);
			for (j = 1; j < 8; j++)
				r[j] = o_twiddle(r[j], j, alpha);
		}

		o_dft8(r);

		j0 = ((i - k) << 3) + k;		/* fft.cl:314-329 */
		for (j = 0; j < 8; j++)
			dst[j0 + j * p] = r[perm[j]];
	}
}

/* fft.cl:213-273, generalised like radix-8 above (work-group size N/4) */
static void o_pass_radix4(const cf *src, cf *dst, int n, int p)
{
	const int t = n >> 2;
	int i;
	for (i = 0; i < t; i++) {
		cf r[4];
		int k = i & (p - 1);
		int j0 = ((i - k) << 2) + k;
		float alpha = -ORACLE_PI_F * (float)k / (float)(2 * p);
		r[0] = src[i]; r[1] = src[i + t]; r[2] = src[i + 2 * t]; r[3] = src[i + 3 * t];
		r[1] = o_twiddle(r[1], 1, alpha);
		r[2] = o_twiddle(r[2], 2, alpha);
		r[3] = o_twiddle(r[3], 3, alpha);
		o_dft4(r);
		dst[j0] = r[0]; dst[j0 + p] = r[2]; dst[j0 + 2 * p] = r[1]; dst[j0 + 3 * p] = r[3];
	}
}

/* ---- the radix-16 plan of N = 65536 (and 8192) --------------------------------------------------------------
 * The reference has one FFT length, 1024 (fft.cl:397-466), and no butterfly wider than 8.  For the lengths BASELINE
 * configs C3 / C5 name, no reference behaviour exists; the plan below is THIS BUILD'S OWN CHOICE,
 * made for the GPU's memory hierarchy -- two radix-16 passes are a 256-point transform that one wavefront does on its
 * own -- and restated here operation for operation so that the GPU kernels can be checked bit for bit.  It is built
 * from the reference's pieces only: dft2 / dft8 (fft.cl:86-145), the constant rotations mul_p1q2 / p1q4 / p3q4
 * (fft.cl:77-82), twiddle() with the reference's expression for the angle (fft.cl:61-69, 286-297), and the Stockham
 * indexing of fft.cl:278-350 with 16 in the place of 8.
 *
 * dft16: decimation in frequency, one radix-2 stage in front of two dft8:
 *     a[j] = r[j] + r[j + 8],  b[j] = (r[j] - r[j + 8]) * W16^j,  j < 8;   X[2m] = DFT8(a)[m],  X[2m + 1] = DFT8(b)[m]
 * W16^j = twiddle(., j, -pi/8) for odd j (full complex products with the pinned sin / cos), the reference's constant
 * rotations for j = 2, 4, 6.  Like o_dft8 it leaves X[jj] in r[bitrev4(jj)]. */
static inline void o_dft16(cf *r)
{
	const float a16 = -ORACLE_PI_F / 8.0f;
	int j;
	for (j = 0; j < 8; j++)
		o_dft2(&r[j], &r[j + 8]);
	r[9]  = o_twiddle(r[9], 1, a16);
	r[10] = o_mul_p1q4(r[10]);
	r[11] = o_twiddle(r[11], 3, a16);
	r[12] = o_mul_p1q2(r[12]);
	r[13] = o_twiddle(r[13], 5, a16);
	r[14] = o_mul_p3q4(r[14]);
	r[15] = o_twiddle(r[15], 7, a16);
	o_dft8(r);
	o_dft8(r + 8);
}

/* One Stockham radix-16 pass: fft.cl:278-350 with t = N/16 work-items of 16 points, angle -pi k / (8 p). */
static void o_pass_radix16(const cf *src, cf *dst, int n, int p, int tw)
{
	const int t = n >> 4;
	/* X[jj] sits in r[8 (jj & 1) + perm8[jj >> 1]] = r[bitrev4(jj)] */
	static const int perm[16] = { 0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15 };
	int i, j;

	for (i = 0; i < t; i++) {
		cf r[16];
		int k = i & (p - 1);
		int j0;

		for (j = 0; j < 16; j++)
			r[j] = src[i + j * t];

		if (tw) {
			float alpha = -ORACLE_PI_F * (float)k / (float)(8 * p);
			for (j = 1; j < 16; j++)
				r[j] = o_twiddle(r[j], j, alpha);
		}

		o_dft16(r);

		j0 = ((i - k) << 4) + k;
		for (j = 0; j < 16; j++)
			dst[j0 + j * p] = r[perm[j]];
	}
}

/* Final radix-2 pass: fft.cl:428-458 (p = N/2, k = i, t = N/2) */
static void o_pass_radix2(const cf *src, cf *dst, int n)
{
	const int p = n >> 1;
	int i;
	for (i = 0; i < p; i++) {
		cf r0 = src[i], r1 = src[i + p];	/* fft.cl:169-176 */
		int k = i & (p - 1);
		float alpha = -ORACLE_PI_F * (float)k / (float)(p);	/* fft.cl:161-167 */
		int j0 = ((i - k) << 1) + k;		/* fft.cl:178-187 */
		r1 = o_twiddle(r1, 1, alpha);
		o_dft2(&r0, &r1);
		dst[j0] = r0;
		dst[j0 + p] = r1;
	}
}

/* fft.cl:397-466 (N=1024) and fft.cl:357-394 (N=512); other N by the same
 * plan: as many radix-8 passes as fit, then radix-4 or radix-2. */
static void o_fft_one(int log2n, const cf *in, cf *out, const float *win, cf *scratch)
{
	const int n = 1 << log2n;
	cf *a = scratch, *b = scratch + n, *tmp;
	int i, p, done;

	for (i = 0; i < n; i++) {			/* fft.cl:415-417 */
		a[i].x = in[i].x * win[i];
		a[i].y = in[i].y * win[i];
	}

	p = 1;
	if (log2n == 16) {
		/* this build's plan for N = 65536: four radix-16 passes, p = 1, 16, 256, 4096 (see o_dft16) */
		for (done = 0; done < 16; done += 4) {
			o_pass_radix16(a, b, n, p, p > 1);
			tmp = a; a = b; b = tmp;
			p <<= 4;
		}
		memcpy(out, a, sizeof(cf) * (size_t)n);
		return;
	}
	for (done = 0; done + 3 <= log2n; done += 3) {
		o_pass_radix8(a, b, n, p, p > 1);
		tmp = a; a = b; b = tmp;
		p <<= 3;
	}
	if (log2n - done == 2) {
		o_pass_radix4(a, b, n, p);
		tmp = a; a = b; b = tmp;
	} else if (log2n - done == 1) {
		o_pass_radix2(a, b, n);
		tmp = a; a = b; b = tmp;
	}

	memcpy(out, a, sizeof(cf) * (size_t)n);		/* fft.cl:460-462 */
}

void fosphor_oracle_fft(int log2n, const float *in, float *out, const float *win, int n_spectra)
{
	const int n = 1 << log2n;
	cf *scratch = (cf *)malloc(sizeof(cf) * 2 * (size_t)n);
	int s;
	for (s = 0; s < n_spectra; s++)
		o_fft_one(log2n, (const cf *)in + (size_t)s * n, (cf *)out + (size_t)s * n, win, scratch);
	free(scratch);
}

/* ------------------------------------------------------------------------ */
/* State                                                                    */
/* ------------------------------------------------------------------------ */

struct fosphor_oracle {
	int log2n, n, n_bins, wf_rows;

	float *win;		/* [N] */
	float *wf;		/* [wf_rows][N] */
	float *hist;		/* [n_bins][N] */
	float *spectrum;	/* [2][N][2] */
	uint32_t *hc;		/* [N][n_bins], last call */
	float *fft_out;		/* last call */
	size_t fft_out_cap;

	int booted;		/* cl.c:92-96 CL_BOOTING vs later */
	int wf_pos;		/* cl.c:954 */

	float pwr_scale, pwr_offset;		/* fosphor.c:147-150 */
	float histo_scale, histo_offset;	/* cl.c:1087-1088 */
	float t0r, t0d, alpha;			/* cl.c:714-716 */
};

fosphor_oracle *fosphor_oracle_new(int log2n, int n_bins, int wf_rows)
{
	fosphor_oracle *st = (fosphor_oracle *)calloc(1, sizeof(*st));
	if (!st) return NULL;
	st->log2n = log2n; st->n = 1 << log2n; st->n_bins = n_bins; st->wf_rows = wf_rows;
	st->win      = (float *)calloc((size_t)st->n, sizeof(float));
	st->wf       = (float *)calloc((size_t)st->n * wf_rows, sizeof(float));
	st->hist     = (float *)calloc((size_t)st->n * n_bins, sizeof(float));
	st->spectrum = (float *)calloc((size_t)st->n * 4, sizeof(float));
	st->hc       = (uint32_t *)calloc((size_t)st->n * n_bins, sizeof(uint32_t));
	st->t0r = 16.0f; st->t0d = 1024.0f; st->alpha = 0.002f;
	fosphor_oracle_set_window_default(st);
	fosphor_oracle_set_power_range(st, 0, 10);	/* fosphor.c:64-66 */
	return st;
}

void fosphor_oracle_free(fosphor_oracle *st)
{
	if (!st) return;
	free(st->win); free(st->wf); free(st->hist); free(st->spectrum); free(st->hc); free(st->fft_out);
	free(st);
}

/* fosphor.c:108-121 -- cosf here is glibc's: the window is an INPUT to the
 * compute core (the product takes the same array), not part of the pinned path */
void fosphor_oracle_set_window_default(fosphor_oracle *st)
{
	int i;
	for (i = 0; i < st->n; i++) {
		float ft = (float)st->n;
		float fp = (float)i;
		st->win[i] = (0.54f - 0.46f * cosf((2.0f * 3.141592f * fp) / ft)) * 1.855f;
	}
}

void fosphor_oracle_set_window(fosphor_oracle *st, const float *win)
{
	memcpy(st->win, win, sizeof(float) * (size_t)st->n);
}

/* fosphor.c:131-152, cl.c:1081-1089.  log10f(N) via the pinned primitive
 * (equal to glibc's for N = 2^k, checked in tests). */
void fosphor_oracle_set_power_range(fosphor_oracle *st, int db_ref, int db_per_div)
{
	int db0 = db_ref - 10 * db_per_div;
	int db1 = db_ref;
	float k = fpm_log10f((float)st->n);
	st->pwr_offset = -(k + ((float)db0 / 20.0f));
	st->pwr_scale  = 20.0f / (float)(db1 - db0);
	st->histo_scale  = st->pwr_scale * (float)st->n_bins;
	st->histo_offset = st->pwr_offset;
}

void fosphor_oracle_set_constants(fosphor_oracle *st, float t0r, float t0d, float alpha)
{
	st->t0r = t0r; st->t0d = t0d; st->alpha = alpha;
}

/* ------------------------------------------------------------------------ */
/* Display (display.cl)                                                     */
/* ------------------------------------------------------------------------ */

/* OpenCL max(): "y if x < y, otherwise x" */
static inline float o_max(float x, float y) { return (x < y) ? y : x; }
/* OpenCL clamp(): min(max(x, lo), hi) with min(a,b) = "b if b < a, otherwise a" */
static inline float o_clamp(float x, float lo, float hi)
{
	float t = (x < lo) ? lo : x;
	return (hi < t) ? hi : t;
}
/* native_powr / native_recip bindings (feed tolerance-checked floats only) */
static inline float o_powr(float x, float y) { return powf(x, y); }
static inline float o_recip(float x) { return 1.0f / x; }

int fosphor_oracle_bin(float re, float im, float hs, float ho, int n_bins)
{
	float pwr = fpm_log10f(fpm_hypotf(re, im));	/* display.cl:136 */
	return fpm_bin_from_pwr(pwr, hs, ho, n_bins);	/* display.cl:161-168 */
}

/* The twiddle factor the restatement multiplies by, for table checks:
 * radix 8 (fft.cl:285-297): alpha = -pi*k/(4p), factor n; radix 2 (fft.cl:161-167): radix2 != 0,
 * alpha = -pi*k/p, n = 1. */
void fosphor_oracle_twiddle(int radix2, int p, int k, int n, float *cs)
{
	float alpha = radix2 ? (-ORACLE_PI_F * (float)k / (float)(p))
	                     : (-ORACLE_PI_F * (float)k / (float)(4 * p));
	cs[0] = fpm_cosf((float)n * alpha);
	cs[1] = fpm_sinf((float)n * alpha);
}

/* Vectorised form for kernel-level tests: bin index and log-power of n FFT outputs */
void fosphor_oracle_bins(const float *fft, int n, float hs, float ho, int n_bins, int32_t *bin, float *pwr)
{
	int i;
	for (i = 0; i < n; i++) {
		float p = fpm_log10f(fpm_hypotf(fft[2 * i], fft[2 * i + 1]));
		pwr[i] = p;
		bin[i] = fpm_bin_from_pwr(p, hs, ho, n_bins);
	}
}

/* One display work-group = 16 consecutive columns (display.cl:67, cl.c:945-950).
 * x0 = first column.  Everything below follows display.cl line by line with
 * the 16x16 local geometry kept, because the float summation orders of
 * live_buf (display.cl:149-150,196-197) depend on it. */
static void o_display_group(fosphor_oracle *st, const cf *fft, int batch, int wf_offset, int x0)
{
	const int n = st->n, nb = st->n_bins;
	const float oma = 1.0f - st->alpha;	/* display.cl:99 */
	float live_buf[16][16];			/* [l1][l0], display.cl:94 */
	float max_buf[16][16];			/* display.cl:95 */
	uint32_t *histo = (uint32_t *)calloc((size_t)nb * 16, sizeof(uint32_t));	/* [bin][l0], display.cl:96 */
	int l0, l1, gidx, b;

	for (l1 = 0; l1 < 16; l1++)
		for (l0 = 0; l0 < 16; l0++) {
			float max_pwr = -1000.0f;	/* display.cl:91 */
			float acc = 0.0f;		/* display.cl:113 */
			int x = x0 + l0;

			for (gidx = 0; gidx < batch; gidx += 16) {	/* display.cl:130 */
				int t = gidx + l1;
				cf v = fft[(size_t)t * n + x];		/* display.cl:133-134 */
				float pwr = fpm_log10f(fpm_hypotf(v.x, v.y));	/* :136 */
				int bin;

				max_pwr = o_max(max_pwr, pwr);		/* :139 */

				st->wf[(size_t)((t + wf_offset) & (st->wf_rows - 1)) * n + x] = pwr;	/* :142-146 */

				acc += pwr * o_powr(oma, (float)(batch - gidx - l1 - 1));	/* :149-150 */

				bin = fpm_bin_from_pwr(pwr, st->histo_scale, st->histo_offset, nb);	/* :161-168 */
				histo[bin * 16 + l0]++;			/* :176 */
			}
			live_buf[l1][l0] = acc;
			max_buf[l1][l0] = max_pwr;	/* :180 */
		}

	for (l0 = 0; l0 < 16; l0++) {
		int x = x0 + l0;
		int half = n >> 1;
		int i = x ^ half;		/* :200-201 */
		float sum = 0.0f, vy, m;
		int j;

		/* Live spectrum, display.cl:188-214 */
		for (j = 0; j < 16; j++)
			sum += live_buf[j][l0];
		vy = st->spectrum[2 * i + 1];
		if (!isfinite(vy))
			vy = sum / 16.0f;
		vy = vy * o_powr(oma, (float)batch) + sum * st->alpha;
		st->spectrum[2 * i + 0] = ((float)i / (float)half) - 1.0f;
		st->spectrum[2 * i + 1] = vy;

		/* Histogram rise/decay, display.cl:217-254 */
		for (b = 0; b < nb; b++) {
			float hv = st->hist[(size_t)b * n + x];
			uint32_t hc = histo[b * 16 + l0];
			float fa, fb, fc, fd, fe;

			st->hc[(size_t)x * nb + b] = hc;

			if ((hv <= 0.01f) && (hc == 0))		/* :237-238 */
				continue;

			fa = (float)hc / (float)batch;		/* :241-245 */
			fb = fa * o_recip(st->t0r);
			fc = fb + o_recip(st->t0d);
			fd = fb * o_recip(fc);
			fe = o_powr(1.0f - fc, (float)batch);

			hv = (hv - fd) * fe + fd;		/* :247 */
			hv = o_clamp(hv, 0.0f, 1.0f);		/* :250 */
			st->hist[(size_t)b * n + x] = hv;
		}

		/* Max hold with decay, display.cl:257-310 (MAX_HOLD_DECAY) */
		m = st->spectrum[2 * (n + i) + 1];
		if (!isfinite(m))
			m = -3.402823466e+38f;			/* -MAXFLOAT, :290-291 */
		m = m * 0.999f + 0.001f * st->spectrum[2 * i + 1];	/* :303, uses the updated live value */
		for (j = 0; j < 16; j++)
			m = o_max(m, max_buf[j][l0]);		/* :304-305 */
		st->spectrum[2 * (n + i) + 0] = ((float)i / (float)half) - 1.0f;
		st->spectrum[2 * (n + i) + 1] = m;
	}

	free(histo);
}

/* ------------------------------------------------------------------------ */
/* Process  (cl.c:870-968)                                                  */
/* ------------------------------------------------------------------------ */

struct o_job {
	fosphor_oracle *st;
	const cf *in;
	int batch, wf_offset;
	int tid, nthreads;
	int phase;
};

static void *o_worker(void *arg)
{
	struct o_job *j = (struct o_job *)arg;
	fosphor_oracle *st = j->st;
	const int n = st->n;

	if (j->phase == 0) {
		cf *scratch = (cf *)malloc(sizeof(cf) * 2 * (size_t)n);
		int s;
		for (s = j->tid; s < j->batch; s += j->nthreads)
			o_fft_one(st->log2n, j->in + (size_t)s * n, (cf *)st->fft_out + (size_t)s * n, st->win, scratch);
		free(scratch);
	} else {
		int g;
		for (g = j->tid; g < n / 16; g += j->nthreads)
			o_display_group(st, (const cf *)st->fft_out, j->batch, j->wf_offset, g * 16);
	}
	return NULL;
}

static void o_run(fosphor_oracle *st, const cf *in, int batch, int phase, int nthreads)
{
	enum { MAXT = 256 };
	pthread_t th[MAXT];
	struct o_job jobs[MAXT];
	int t;

	if (nthreads < 1) nthreads = 1;
	if (nthreads > MAXT) nthreads = MAXT;
	for (t = 0; t < nthreads; t++) {
		jobs[t].st = st; jobs[t].in = in; jobs[t].batch = batch; jobs[t].wf_offset = st->wf_pos;
		jobs[t].tid = t; jobs[t].nthreads = nthreads; jobs[t].phase = phase;
	}
	if (nthreads == 1) {
		o_worker(&jobs[0]);
		return;
	}
	for (t = 0; t < nthreads; t++)
		pthread_create(&th[t], NULL, o_worker, &jobs[t]);
	for (t = 0; t < nthreads; t++)
		pthread_join(th[t], NULL);
}

int fosphor_oracle_process(fosphor_oracle *st, const float *samples, int len, int strict, int nthreads)
{
	const int n = st->n;
	int batch;
	size_t need, i;

	if (len <= 0 || (len & ((16 * n) - 1)))		/* cl.c:882-883 */
		return -EINVAL;
	if (strict && len > n * 1024)			/* cl.c:885-886 */
		return -EINVAL;
	batch = len / n;

	need = (size_t)len * 2;
	if (need > st->fft_out_cap) {
		free(st->fft_out);
		st->fft_out = (float *)malloc(need * sizeof(float));
		st->fft_out_cap = need;
	}

	o_run(st, (const cf *)samples, batch, 0, nthreads);	/* cl.c:913-920 */

	if (!st->booted) {				/* cl.c:930-934, 406-465 */
		float noise_floor = -st->pwr_offset;
		for (i = 0; i < (size_t)n * 4; i++) st->spectrum[i] = noise_floor;
		for (i = 0; i < (size_t)n * st->wf_rows; i++) st->wf[i] = noise_floor;
		for (i = 0; i < (size_t)n * st->n_bins; i++) st->hist[i] = 0.0f;
		st->booted = 1;
	}

	o_run(st, NULL, batch, 1, nthreads);		/* cl.c:937-951 */

	st->wf_pos = (st->wf_pos + batch) & (st->wf_rows - 1);	/* cl.c:954 */
	return 0;
}

float    *fosphor_oracle_waterfall(fosphor_oracle *st) { return st->wf; }
float    *fosphor_oracle_histogram(fosphor_oracle *st) { return st->hist; }
float    *fosphor_oracle_spectrum(fosphor_oracle *st)  { return st->spectrum; }
uint32_t *fosphor_oracle_hitcount(fosphor_oracle *st)  { return st->hc; }
float    *fosphor_oracle_fft_out(fosphor_oracle *st)   { return st->fft_out; }
int       fosphor_oracle_waterfall_pos(fosphor_oracle *st) { return st->wf_pos; }
float     fosphor_oracle_histo_scale(fosphor_oracle *st)  { return st->histo_scale; }
float     fosphor_oracle_histo_offset(fosphor_oracle *st) { return st->histo_offset; }
const float *fosphor_oracle_window(fosphor_oracle *st) { return st->win; }
