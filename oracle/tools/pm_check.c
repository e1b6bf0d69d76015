/*
 * pm_check.c -- offline validation of include/fosphor_portable_math.h
 *
 * TEST INFRASTRUCTURE (oracle side).  Not linked into the product.
 *
 *   pm_check quick        sampled checks, < 2 s      (run by tests/, CPU)
 *   pm_check exhaustive   every positive float through log10f / bin
 *                         monotonicity, every |v| < 2^24 through roundf
 *                         (about a minute on 8 cores; run by hand, result
 *                         recorded in DESIGN.md)
 *
 * Checks:
 *   1. fpm_log10f is monotone non-decreasing over all positive floats and
 *      within 1 ulp of (float)log10((double)h) (glibc's log10f itself is up to 2 ulp off).
 *   2. fpm_roundf == glibc roundf (half away from zero).
 *   3. fpm_hypotf vs glibc hypotf: mismatch count on random pairs.
 *   4. fpm_sinf / fpm_cosf vs (float)sin/cos(double): mismatch count on the
 *      FFT's actual twiddle arguments and on a dense sweep.
 *   5. fpm_bin_from_sqmag monotone in s for the default power range.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

#include "../../include/fosphor_portable_math.h"

static int g_fail;

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint64_t rng(void)
{
	uint64_t z = (rng_state += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

static int ulp_diff(float a, float b)
{
	int32_t ia, ib;
	if (a == b) return 0;
	if (a != a || b != b) return (a != a && b != b) ? 0 : 1 << 30;
	memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
	if (ia < 0) ia = (int32_t)0x80000000 - ia;
	if (ib < 0) ib = (int32_t)0x80000000 - ib;
	return abs(ia - ib);
}

struct span { uint32_t lo, hi; int bad_mono; int max_ulp; long n_diff; };

static void *log_span(void *arg)
{
	struct span *s = (struct span *)arg;
	float prev = fpm_log10f(fpm_u2f(s->lo));
	uint32_t u;
	for (u = s->lo; u < s->hi; u++) {
		float h = fpm_u2f(u);
		float l = fpm_log10f(h);
		int d = ulp_diff(l, (float)log10((double)h));
		if (l < prev) s->bad_mono++;
		if (d > s->max_ulp) s->max_ulp = d;
		if (d) s->n_diff++;
		prev = l;
	}
	return NULL;
}

static void check_log10(int exhaustive)
{
	enum { NT = 8 };
	pthread_t th[NT];
	struct span sp[NT];
	uint32_t lo = 1, hi = 0x7f800000u;	/* all positive denormal+normal floats */
	int i, bad = 0, mu = 0; long nd = 0;

	if (!exhaustive) {
		/* sampled: 2^21 floats per exponent-octave-ish stride */
		uint32_t u; float prev = -INFINITY;
		for (u = 1; u < 0x7f800000u; u += 1021) {
			float l = fpm_log10f(fpm_u2f(u));
			int d = ulp_diff(l, (float)log10((double)fpm_u2f(u)));
			if (l < prev) bad++;
			if (d > mu) mu = d;
			if (d) nd++;
			prev = l;
		}
	} else {
		for (i = 0; i < NT; i++) {
			uint64_t a = lo + (uint64_t)(hi - lo) * i / NT;
			uint64_t b = lo + (uint64_t)(hi - lo) * (i + 1) / NT;
			/* overlap by one so span boundaries are compared too */
			sp[i].lo = (uint32_t)(i ? a - 1 : a); sp[i].hi = (uint32_t)b;
			sp[i].bad_mono = 0; sp[i].max_ulp = 0; sp[i].n_diff = 0;
			pthread_create(&th[i], NULL, log_span, &sp[i]);
		}
		for (i = 0; i < NT; i++) {
			pthread_join(th[i], NULL);
			bad += sp[i].bad_mono;
			if (sp[i].max_ulp > mu) mu = sp[i].max_ulp;
			nd += sp[i].n_diff;
		}
	}
	printf("log10f: monotonicity violations %d, max ulp vs (float)glibc-log10(double) %d, differing inputs %ld (%s)\n",
	       bad, mu, nd, exhaustive ? "exhaustive" : "sampled");
	if (bad || mu > 1) g_fail = 1;
	if (fpm_log10f(0.0f) != -INFINITY || fpm_log10f(INFINITY) != INFINITY ||
	    fpm_log10f(1.0f) != 0.0f || fpm_log10f(10.0f) != 1.0f ||
	    fpm_log10f(-1.0f) == fpm_log10f(-1.0f)) {
		printf("log10f: special value failure\n");
		g_fail = 1;
	}
}

static void check_round(int exhaustive)
{
	long bad = 0; uint32_t u, step = exhaustive ? 1 : 257;
	for (u = 0; u < 0x4c000000u; u += step) {	/* |v| < 2^25 */
		float v = fpm_u2f(u);
		if (fpm_f2u(fpm_roundf(v)) != fpm_f2u(roundf(v))) bad++;
		if (fpm_f2u(fpm_roundf(-v)) != fpm_f2u(roundf(-v))) bad++;
	}
	if (fpm_roundf(INFINITY) != INFINITY || fpm_roundf(-INFINITY) != -INFINITY) bad++;
	if (fpm_roundf(2.5f) != 3.0f || fpm_roundf(-2.5f) != -3.0f || fpm_roundf(0.49999997f) != 0.0f) bad++;
	printf("roundf: mismatches vs glibc %ld\n", bad);
	if (bad) g_fail = 1;
}

static void check_hypot(int exhaustive)
{
	long n = exhaustive ? 200000000L : 2000000L, i, bad = 0;
	for (i = 0; i < n; i++) {
		uint64_t r = rng();
		/* exponent range 2^-40 .. 2^40, random mantissas and signs */
		uint32_t ua = ((uint32_t)r & 0x807fffffu) | ((87u + (uint32_t)((r >> 32) % 80)) << 23);
		uint32_t ub = ((uint32_t)(r >> 20) & 0x807fffffu) | ((87u + (uint32_t)((r >> 40) % 80)) << 23);
		float a = fpm_u2f(ua), b = fpm_u2f(ub);
		if (fpm_f2u(fpm_hypotf(a, b)) != fpm_f2u(hypotf(a, b))) bad++;
	}
	printf("hypotf: mismatches vs glibc %ld of %ld random pairs\n", bad, n);
	/* informational: glibc is not the definition; but gross disagreement is a bug */
	if (bad > n / 1000) g_fail = 1;
	if (fpm_hypotf(INFINITY, NAN) != INFINITY || fpm_hypotf(3.0f, 4.0f) != 5.0f ||
	    fpm_hypotf(0.0f, 0.0f) != 0.0f) { printf("hypotf: special value failure\n"); g_fail = 1; }
}

static void check_sincos(void)
{
	long bad = 0, n = 0; int p, k, m;
	const float M_PIf = 3.141592653589f;
	/* the exact arguments formed by fft.cl:286-297 and fft.cl:162-166 */
	for (p = 8; p <= 4096; p *= 8)
		for (k = 0; k < p; k++) {
			float alpha = -M_PIf * (float)k / (float)(4 * p);
			for (m = 1; m < 8; m++) {
				float a = m * alpha;
				if (fpm_f2u(fpm_sinf(a)) != fpm_f2u((float)sin((double)a))) bad++;
				if (fpm_f2u(fpm_cosf(a)) != fpm_f2u((float)cos((double)a))) bad++;
				n += 2;
			}
		}
	for (p = 512; p <= 32768; p *= 8)
		for (k = 0; k < p; k++) {
			float a = -M_PIf * (float)k / (float)p;
			if (fpm_f2u(fpm_sinf(a)) != fpm_f2u((float)sin((double)a))) bad++;
			if (fpm_f2u(fpm_cosf(a)) != fpm_f2u((float)cos((double)a))) bad++;
			n += 2;
		}
	for (k = -700000; k <= 700000; k++) {
		float a = (float)k * 1e-5f;
		if (ulp_diff(fpm_sinf(a), (float)sin((double)a)) > 1) bad += 1000;
		if (ulp_diff(fpm_cosf(a), (float)cos((double)a)) > 1) bad += 1000;
	}
	printf("sin/cos: mismatches vs (float)glibc-double on %ld twiddle args: %ld\n", n, bad);
	if (bad) g_fail = 1;
	if (fpm_sinf(0.0f) != 0.0f || fpm_cosf(0.0f) != 1.0f) g_fail = 1;
}

static void check_bin_monotone(int exhaustive)
{
	/* default power range: scale 0.2*128, offset 1.9896998 (fosphor.c:131-152) */
	const float ofs = -(fpm_log10f(1024.0f) + (-100.0f / 20.0f));
	int nb, bad = 0;
	for (nb = 128; nb <= 512; nb *= 2) {
		float hs = (20.0f / 100.0f) * (float)nb;
		uint32_t u, step = exhaustive ? 1 : 509; int prev = 0;
		for (u = 0; u < 0x7f800000u; u += step) {
			float h = fpm_u2f(u);		/* h sweeps all floats; s = h^2 exactly in double */
			int b = fpm_bin_from_sqmag((double)h * (double)h, hs, ofs, nb);
			if (b < prev) bad++;
			prev = b;
		}
	}
	printf("bin(s): monotonicity violations %d\n", bad);
	if (bad) g_fail = 1;
}

int main(int argc, char **argv)
{
	int ex = (argc > 1 && !strcmp(argv[1], "exhaustive"));
	check_log10(ex);
	check_round(ex);
	check_hypot(ex);
	check_sincos();
	check_bin_monotone(ex);
	printf(g_fail ? "PM_CHECK FAIL\n" : "PM_CHECK OK\n");
	return g_fail;
}
