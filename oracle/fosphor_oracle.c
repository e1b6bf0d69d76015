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

/* ---- the plans of the long lengths: N = 8192 (BASELINE C3) and N = 65536 (C5) -----------------------------------
 * The reference has ONE FFT length, 1024 (fft.cl:397-466), and no reference behaviour exists for the lengths BASELINE
 * configs C3 / C5 name.  The plans below are THIS BUILD'S OWN CHOICE, restated here operation for
 * operation so that the GPU kernels can be checked bit for bit.  They keep the reference's data flow -- Stockham passes with
 * the indexing of fft.cl:278-350 with 16 in the place of 8: radix 16, p = 1, 16, 256 and then the radix-2 pass of fft.cl:428-458
 * (p = 4096) at N = 8192 (sixteen points per work-item: the GPU kernel makes two exchanges and a half through its LDS per spectrum);
 * p = 1, 16, 256, 4096 at N = 65536 --, twiddle angles from the reference's expression
 * -pi k / ((R/2) p) (fft.cl:286-297) through the pinned sin / cos -- and change the ARITHMETIC INSIDE A PASS to what a
 * fused-multiply-add machine does best: the radix-R butterfly is log2(R) radix-2 stages in decimation-in-time form whose
 * twiddles sit ON the butterflies,
 *         a' = a + T b        two fused multiply-adds per component pair:  u = fma(b.yx, (-T.y, T.y), a);  a' = fma(b, T.x, u)
 *         b' = 2 a - a'       one:                                          b' = fma(2, a, -a')
 * instead of "multiply every input by its twiddle (fft.cl:37-46: 2 mul + 1 add/sub per product), then an untwiddled dft8
 * (fft.cl:112-145)".  Per radix-16 pass that is 32 butterflies x 3 packed operations = 96 per 16 points against 133, with 8
 * twiddles per item (w^8, w^4, w^2, w^2 W8, w, w W16, w W8, w W16^3) instead of 15.
 * Every operation is ONE IEEE fused multiply-add (fmaf here, v_pk_fma_f32 there) or one add / multiply: bit-reproducible.
 *
 * Derivation (R = 16): X[m] = sum_j W16^(j m) w^j r[j].  Splitting j = j' + 8 j1:
 *   X[m] = sum_{j' < 8} W16^(j' m) w^j' ( r[j'] + (-1)^m0 w^8 r[j' + 8] ),  m0 = m & 1        -> stage A, T = w^8
 * and again with distances 4, 2, 1: the twiddle of a stage is w^(R / 2^s) times the power of W16 its position implies,
 *   B: T = w^4 (-j)^m0      C: T = w^2 W8^m0 (-j)^m1      D: T = w W16^m0 W8^m1 (-j)^m2
 * (the '+' output stays where a was, the '-' output goes where b was), so that X[m0 + 2 m1 + 4 m2 + 8 m3] ends in register
 * 8 m0 + 4 m1 + 2 m2 + m3 = bitrev4(m), the order the reference's dft8 leaves its results in as well (fft.cl:321-328).
 * A factor -j is a swap of the components with one sign change and costs nothing (o_bf_mj).
 * The window multiply of fft.cl:415-417 is folded into stage A of the first pass (p = 1, w = 1):
 *   m = a win_a;  a' = fma(b, win_b, m);  b' = fma(b, -win_b, m). */

/* exp(-j pi q / den) through the pinned sin / cos: every twiddle of the long plans */
static inline cf o_tw(int q, int den)
{
	cf w;
	float arg = -ORACLE_PI_F * (float)q / (float)den;
	w.x = fpm_cosf(arg);
	w.y = fpm_sinf(arg);
	return w;
}

/* a' = a + T b, b' = 2 a - a' */
static inline void o_bf(cf *a, cf *b, cf t)
{
	float ux = fmaf(-b->y, t.y, a->x);
	float uy = fmaf( b->x, t.y, a->y);
	float px = fmaf( b->x, t.x, ux);
	float py = fmaf( b->y, t.x, uy);
	b->x = fmaf(a->x, 2.0f, -px);
	b->y = fmaf(a->y, 2.0f, -py);
	a->x = px; a->y = py;
}

/* the same with T := -j T:  (-j)(T b) = (Im(T b), -Re(T b)) */
static inline void o_bf_mj(cf *a, cf *b, cf t)
{
	float ux = fmaf( b->x, t.y, a->x);
	float uy = fmaf( b->y, t.y, a->y);
	float px = fmaf( b->y, t.x, ux);
	float py = fmaf(-b->x, t.x, uy);
	b->x = fmaf(a->x, 2.0f, -px);
	b->y = fmaf(a->y, 2.0f, -py);
	a->x = px; a->y = py;
}

/* T = 1 and T = -j: plain sums (fft.cl:86-94 with and without mul_p1q2) */
static inline void o_bf1_mj(cf *a, cf *b)
{
	cf t;
	t.x = a->x - b->y;  t.y = a->y + b->x;
	a->x = a->x + b->y; a->y = a->y - b->x;
	*b = t;
}

/* stage A of the first pass: the window taps ride on the butterfly */
static inline void o_bf_win(cf *a, cf *b, float wa, float wb)
{
	float mx = a->x * wa, my = a->y * wa;
	a->x = fmaf( b->x, wb, mx);
	a->y = fmaf( b->y, wb, my);
	b->x = fmaf(-b->x, wb, mx);
	b->y = fmaf(-b->y, wb, my);
}

/* One Stockham radix-16 pass of the long plans (indexing as fft.cl:278-350 with 16 for 8): t = N/16 items of 16 points, angle
 * -pi k / (8 p).  win != NULL: the first pass (p = 1). */
static void o_pass_radix16_fma(const cf *src, cf *dst, int n, int p, const float *win)
{
	const int t = n >> 4;
	/* X[m] sits in r[bitrev4(m)] */
	static const int perm[16] = { 0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15 };
	const cf w16 = o_tw(1, 8), w8 = o_tw(2, 8), w163 = o_tw(3, 8);
	int i, j;

	for (i = 0; i < t; i++) {
		cf r[16];
		const int k = i & (p - 1);
		const int j0 = ((i - k) << 4) + k;

		for (j = 0; j < 16; j++)
			r[j] = src[i + j * t];

		if (win) {
			for (j = 0; j < 8; j++)
				o_bf_win(&r[j], &r[j + 8], win[i + j * t], win[i + (j + 8) * t]);
			for (j = 0; j < 4; j++) {
				o_dft2(&r[j], &r[j + 4]);
				o_bf1_mj(&r[8 + j], &r[12 + j]);
			}
			for (j = 0; j < 2; j++) {
				o_dft2(&r[j], &r[j + 2]);
				o_bf1_mj(&r[4 + j], &r[6 + j]);
				o_bf(&r[8 + j], &r[10 + j], w8);
				o_bf_mj(&r[12 + j], &r[14 + j], w8);
			}
			o_dft2(&r[0], &r[1]);       o_bf1_mj(&r[2], &r[3]);       o_bf(&r[4], &r[5], w8);     o_bf_mj(&r[6], &r[7], w8);
			o_bf(&r[8], &r[9], w16);    o_bf_mj(&r[10], &r[11], w16); o_bf(&r[12], &r[13], w163); o_bf_mj(&r[14], &r[15], w163);
		} else {
			const int d = 8 * p;
			const cf t8 = o_tw(8 * k, d), t4 = o_tw(4 * k, d), t2 = o_tw(2 * k, d), t2w = o_tw(2 * k + 2 * p, d);
			const cf t1 = o_tw(k, d), t1a = o_tw(k + p, d), t1b = o_tw(k + 2 * p, d), t1c = o_tw(k + 3 * p, d);
			for (j = 0; j < 8; j++)
				o_bf(&r[j], &r[j + 8], t8);
			for (j = 0; j < 4; j++) {
				o_bf(&r[j], &r[j + 4], t4);
				o_bf_mj(&r[8 + j], &r[12 + j], t4);
			}
			for (j = 0; j < 2; j++) {
				o_bf(&r[j], &r[j + 2], t2);
				o_bf_mj(&r[4 + j], &r[6 + j], t2);
				o_bf(&r[8 + j], &r[10 + j], t2w);
				o_bf_mj(&r[12 + j], &r[14 + j], t2w);
			}
			o_bf(&r[0], &r[1], t1);    o_bf_mj(&r[2], &r[3], t1);     o_bf(&r[4], &r[5], t1b);    o_bf_mj(&r[6], &r[7], t1b);
			o_bf(&r[8], &r[9], t1a);   o_bf_mj(&r[10], &r[11], t1a);  o_bf(&r[12], &r[13], t1c);  o_bf_mj(&r[14], &r[15], t1c);
		}

		for (j = 0; j < 16; j++)
			dst[j0 + j * p] = r[perm[j]];
	}
}

/* The final radix-2 pass of the long plan at N = 8192 (fft.cl:428-458: p = N/2, k = i): one butterfly with T = exp(-j pi k / p) */
static void o_pass_radix2_fma(const cf *src, cf *dst, int n)
{
	const int p = n >> 1;
	int i;
	for (i = 0; i < p; i++) {
		cf r0 = src[i], r1 = src[i + p];
		o_bf(&r0, &r1, o_tw(i, p));
		dst[i] = r0;
		dst[i + p] = r1;
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

	if (log2n == 13 || log2n == 16) {
		/* this build's plans for the long lengths (see o_pass_radix16_fma): radix-16 passes p = 1, 16, 256 (, 4096), the window on the first;
		 * N = 8192 ends with the radix-2 pass p = 4096 */
		const cf *src = in;
		p = 1;
		for (done = 0; done + 4 <= log2n; done += 4) {
			o_pass_radix16_fma(src, b, n, p, p == 1 ? win : NULL);
			tmp = a; a = b; b = tmp;
			src = a;
			p <<= 4;
		}
		if (log2n - done == 1) {
			o_pass_radix2_fma(a, b, n);
			tmp = a; a = b; b = tmp;
		}
		memcpy(out, a, sizeof(cf) * (size_t)n);
		return;
	}

	for (i = 0; i < n; i++) {			/* fft.cl:415-417 */
		a[i].x = in[i].x * win[i];
		a[i].y = in[i].y * win[i];
	}

	p = 1;
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
