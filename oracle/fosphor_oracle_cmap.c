/*
 * fosphor_oracle_cmap.c -- CPU restatement of the reference's colour mapping
 *
 * TEST INFRASTRUCTURE (see fosphor_oracle.h): only tests/ may load it.
 *
 * Restates
 *   lib/fosphor/gl_cmap_gen.c   palette generators (histogram :150-178, waterfall :181-198,
 *                               prog :271-322; HSV->RGB :36-107, byte packing :110-121)
 *   lib/fosphor/cmap_simple.glsl:41-47 + gl_cmap.c:303-316 + gl.c:396-438
 *                               intensity -> (v + offset) * scale -> GL_LINEAR / CLAMP_TO_EDGE
 *                               lookup in a 1-D RGBA8 palette; fft-shifted columns; newest
 *                               waterfall row / highest histogram bin on top.
 *
 * Pinning: the palettes are checked bit for bit against tests/golden/cmap_palettes.npz, which
 * oracle/gen_golden.py produced by calling the reference's own gl_cmap_gen.c (compiled from
 * where it lies into oracle/_ref/libcmap_ref.so).  The lookup has no reference fixture: GL leaves
 * the filter's precision to the implementation and no GL context exists here; this file DEFINES
 * it (float32, one rounding per operation, channel = (uint8)(c0 + f * (c1 - c0) + 0.5)) --
 * parity unpinned for the lookup.
 */
#include <math.h>
#include <stdint.h>

/* gl_cmap_gen.c:36-107 */
static void hsv_to_rgb(float *rgb, float h, float s, float v)
{
	int i;
	float r, g, b, f, p, q, t;

	if (s <= 0.0f) {
		rgb[0] = rgb[1] = rgb[2] = v;
		return;
	}
	h *= 5.0f;
	i = floor(h);
	f = h - i;
	p = v * (1 - s);
	q = v * (1 - s * f);
	t = v * (1 - s * (1 - f));
	switch (i % 6) {
	case 0:  r = v; g = t; b = p; break;
	case 1:  r = q; g = v; b = p; break;
	case 2:  r = p; g = v; b = t; break;
	case 3:  r = p; g = q; b = v; break;
	case 4:  r = t; g = p; b = v; break;
	default: r = v; g = p; b = q; break;
	}
	rgb[0] = r; rgb[1] = g; rgb[2] = b;
}

/* gl_cmap_gen.c:110-121 */
static uint32_t rgba_of(float r, float g, float b)
{
	unsigned char rc = (unsigned char)roundf(r * 255.0f);
	unsigned char gc = (unsigned char)roundf(g * 255.0f);
	unsigned char bc = (unsigned char)roundf(b * 255.0f);
	return (255u << 24) | ((uint32_t)bc << 16) | ((uint32_t)gc << 8) | rc;
}

static uint32_t rgba_of_hsv(float h, float s, float v)
{
	float rgb[3];
	hsv_to_rgb(rgb, h, s, v);
	return rgba_of(rgb[0], rgb[1], rgb[2]);
}

/* which: 0 histogram, 1 waterfall, 2 prog */
int fosphor_oracle_cmap(int which, uint32_t *rgba, int N)
{
	int i;
	if (N < 2)
		return -1;
	if (which == 0) {				/* gl_cmap_gen.c:150-178 */
		int m = N >> 4;
		for (i = 0; i < m; i++) {
			float p = (1.0f * i) / (N - 1);
			rgba[i] = rgba_of_hsv(0.90f, 0.50f, 0.15f + 4.0f * p);
		}
		for (i = m; i < N; i++) {
			float p = (1.0f * i) / (N - 1);
			rgba[i] = rgba_of_hsv(0.80f - p * 0.80f,
			                      1.00f - ((p < 0.85f) ? 0.0f : ((p - 0.85f) * 3.0f)),
			                      0.60f + ((p < 0.40f) ? p : 0.40f));
		}
		return 0;
	}
	if (which == 1) {				/* gl_cmap_gen.c:181-198 */
		for (i = 0; i < N; i++) {
			float p = (1.0f * i) / (N - 1);
			rgba[i] = rgba_of_hsv(0.75f - (p * 0.75f), 1.0f, (p * 0.95f) + 0.05f);
		}
		return 0;
	}
	if (which == 2) {				/* gl_cmap_gen.c:271-322 */
		static const float pos[13] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12 };
		static const float col[13][3] = {
			{ 0.29f, 0.00f, 0.00f }, { 0.46f, 0.00f, 0.00f }, { 0.62f, 0.00f, 0.00f },
			{ 0.78f, 0.00f, 0.00f }, { 1.00f, 0.00f, 0.00f }, { 1.00f, 0.43f, 0.10f },
			{ 1.00f, 1.00f, 0.00f }, { 1.00f, 1.00f, 1.00f }, { 0.11f, 0.56f, 1.00f },
			{ 0.00f, 0.00f, 0.57f }, { 0.00f, 0.00f, 0.31f }, { 0.00f, 0.00f, 0.19f },
			{ 0.00f, 0.00f, 0.12f },
		};
		const int NC = 12;
		for (i = 0; i < N; i++) {
			float rgb[3];
			float p = 1.0f - ((1.0f * i) / (N - 1));
			float ps = p * 9.0f;
			float m;
			int li, j;
			for (li = 0; li < (NC - 1) && pos[li + 1] < ps; li++);
			ps -= pos[li];
			ps /= pos[li + 1] - pos[li];
			m = ps;
			for (j = 0; j < 3; j++)
				rgb[j] = col[li][j] * (1.0f - m) + col[li + 1][j] * m;
			rgba[i] = rgba_of(rgb[0], rgb[1], rgb[2]);
		}
		return 0;
	}
	return -1;
}

/* cmap_simple.glsl:41-47 with the palette sampled as gl_cmap.c:314-316 sets it up
 * (GL_LINEAR, GL_CLAMP_TO_EDGE; texel centres at (i + 0.5) / n) */
static uint32_t lookup(float t, float scale, float offset, const uint32_t *pal, int n)
{
	float m = (t + offset) * scale;
	float u = m * (float)n - 0.5f;
	float fl, f;
	int i0, i1, ch;
	uint32_t out = 0;

	if (u != u) u = -1.0f;
	if (u < -1.0f) u = -1.0f;
	if (u > (float)n) u = (float)n;
	fl = floorf(u);
	f = u - fl;
	i0 = (int)fl; i1 = i0 + 1;
	if (i0 < 0) i0 = 0;
	if (i0 > n - 1) i0 = n - 1;
	if (i1 < 0) i1 = 0;
	if (i1 > n - 1) i1 = n - 1;
	for (ch = 0; ch < 4; ch++) {
		float c0 = (float)((pal[i0] >> (8 * ch)) & 0xffu);
		float c1 = (float)((pal[i1] >> (8 * ch)) & 0xffu);
		float c = c0 + f * (c1 - c0);
		out |= ((uint32_t)(c + 0.5f) & 0xffu) << (8 * ch);
	}
	return out;
}

/* image 0: waterfall ring float[src_rows][n] -> rows x n, output row r = ring row
 * (pos - 1 - r) mod src_rows (gl.c:403-404: the newest row is on top); image 1: histogram
 * float[src_rows][n] -> src_rows x n, output row r = bin src_rows - 1 - r (gl.c:427-428).
 * Output column c = texel column (c + n/2) mod n (gl.c:396-400). */
int fosphor_oracle_colorize(int image, const float *src, int src_rows, int n, int pos,
                            const uint32_t *pal, int pal_n, float scale, float offset,
                            int rows, uint32_t *dst)
{
	int r, c;
	for (r = 0; r < rows; r++) {
		int sr = (image == 0) ? (((pos - 1 - r) % src_rows) + src_rows) % src_rows : (src_rows - 1 - r);
		for (c = 0; c < n; c++) {
			int sc = (c + n / 2) % n;
			dst[(long)r * n + c] = lookup(src[(long)sr * n + sc], scale, offset, pal, pal_n);
		}
	}
	return 0;
}
