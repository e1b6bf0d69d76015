/*
 * fosphor_cmap.hip -- headless colour mapping (include/fosphor_amd_cmap.h)
 *
 * Host: the three palette generators of the reference's GL front end, restated
 * (lib/fosphor/gl_cmap_gen.c:36-121 HSV->RGB and packing, :150-198 histogram / waterfall,
 * :271-322 "prog").  Every expression keeps the reference's float operation order so the tables
 * are bit-identical to fosphor_gl_cmap_*() compiled for x86 (tests/golden/cmap_palettes.npz).
 * Device: one elementwise kernel, 4 pixels per thread (16-byte load, 16-byte store), palette in
 * LDS.  HBM-bound: 4 B read + 4 B written per pixel.
 */
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include "../../include/fosphor_amd.h"
#include "../../include/fosphor_amd_cmap.h"

/* accessors implemented next to struct fosphor (fosphor_api.cpp) */
extern "C" int  fosphor_amd_priv_palette(struct fosphor *self, int n, uint32_t **d_palette);
extern "C" void fosphor_amd_priv_power(struct fosphor *self, float *scale, float *offset);

namespace {

/* (h, s, v) -> packed RGBA.  The hue circle is cut in FIVE sectors (h * 5), not six, exactly as
 * gl_cmap_gen.c:54 does -- the palettes only use h <= 0.9, and parity is with what the reference
 * draws, not with textbook HSV. */
uint32_t pack_hsv(float h, float s, float v)
{
	float r, g, b;
	if (s <= 0.0f) {
		r = g = b = v;					/* gl_cmap_gen.c:48-52 */
	} else {
		const float hs = h * 5.0f;
		const int   sector = (int)floor(hs);		/* double floor of a float, :55 */
		const float frac = hs - sector;
		const float lo  = v * (1 - s);			/* :57-59, float arithmetic */
		const float dn  = v * (1 - s * frac);
		const float up  = v * (1 - s * (1 - frac));
		/* channel triples per sector, :61-93 (sector 5 and anything else share one row) */
		const float tab[6][3] = {
			{ v, up, lo }, { dn, v, lo }, { lo, v, up }, { lo, dn, v }, { up, lo, v }, { v, lo, dn },
		};
		int k = sector % 6;
		if (k < 0) k = 5;				/* `default:` of the switch */
		r = tab[k][0]; g = tab[k][1]; b = tab[k][2];
	}
	const uint32_t rc = (unsigned char)roundf(r * 255.0f);	/* :115-118 */
	const uint32_t gc = (unsigned char)roundf(g * 255.0f);
	const uint32_t bc = (unsigned char)roundf(b * 255.0f);
	return (255u << 24) | (bc << 16) | (gc << 8) | rc;
}

uint32_t pack_rgb(float r, float g, float b)
{
	const uint32_t rc = (unsigned char)roundf(r * 255.0f);
	const uint32_t gc = (unsigned char)roundf(g * 255.0f);
	const uint32_t bc = (unsigned char)roundf(b * 255.0f);
	return (255u << 24) | (bc << 16) | (gc << 8) | rc;
}

void palette_histogram(uint32_t *rgba, int n)
{
	const int dark = n >> 4;				/* the lowest 1/16: a dim violet ramp, :153-166 */
	for (int i = 0; i < n; i++) {
		const float p = (1.0f * i) / (n - 1);
		if (i < dark) {
			rgba[i] = pack_hsv(0.90f, 0.50f, 0.15f + 4.0f * p);
		} else {						/* :168-177 */
			const float sat = 1.00f - ((p < 0.85f) ? 0.0f : ((p - 0.85f) * 3.0f));
			const float val = 0.60f + ((p < 0.40f) ? p : 0.40f);
			rgba[i] = pack_hsv(0.80f - p * 0.80f, sat, val);
		}
	}
}

void palette_waterfall(uint32_t *rgba, int n)
{
	for (int i = 0; i < n; i++) {				/* :186-195 */
		const float p = (1.0f * i) / (n - 1);
		rgba[i] = pack_hsv(0.75f - (p * 0.75f), 1.0f, (p * 0.95f) + 0.05f);
	}
}

void palette_prog(uint32_t *rgba, int n)
{
	/* 13 colour stops at integer positions 0..12, :274-291; intensity 1 maps to stop 0 and
	 * intensity 0 to position 9 (:299-300), so the last three stops are never reached */
	static const float stop[13][3] = {
		{ 0.29f, 0.00f, 0.00f }, { 0.46f, 0.00f, 0.00f }, { 0.62f, 0.00f, 0.00f }, { 0.78f, 0.00f, 0.00f },
		{ 1.00f, 0.00f, 0.00f }, { 1.00f, 0.43f, 0.10f }, { 1.00f, 1.00f, 0.00f }, { 1.00f, 1.00f, 1.00f },
		{ 0.11f, 0.56f, 1.00f }, { 0.00f, 0.00f, 0.57f }, { 0.00f, 0.00f, 0.31f }, { 0.00f, 0.00f, 0.19f },
		{ 0.00f, 0.00f, 0.12f },
	};
	for (int i = 0; i < n; i++) {
		const float p = 1.0f - ((1.0f * i) / (n - 1));
		float ps = p * 9.0f;
		int li = 0;
		while (li < 11 && (float)(li + 1) < ps)		/* :305-306 with colors[k].p == k */
			li++;
		ps -= (float)li;
		ps /= (float)(li + 1) - (float)li;			/* :309-310 */
		const float m = ps;
		float rgb[3];
		for (int j = 0; j < 3; j++)
			rgb[j] = stop[li][j] * (1.0f - m) + stop[li + 1][j] * m;	/* :315-316 */
		rgba[i] = pack_rgb(rgb[0], rgb[1], rgb[2]);
	}
}

struct CmapParams {
	const float *src;		/* [src_rows][n] */
	uint32_t    *dst;		/* [rows][n] */
	const uint32_t *pal;		/* [pal_n] device */
	int   n, rows;
	int   row_base, row_mask;	/* source row of output row r: (row_base - r) & row_mask */
	int   pal_n;
	float scale, offset;
};

constexpr int kPalMax = 4096;

__device__ __forceinline__ uint32_t lookup(float t, const CmapParams &p, const uint32_t *pal)
{
	const float m = (t + p.offset) * p.scale;		/* cmap_simple.glsl:44 */
	float u = m * (float)p.pal_n - 0.5f;
	u = (u != u) ? -1.0f : u;				/* NaN -> entry 0 */
	u = fminf(fmaxf(u, -1.0f), (float)p.pal_n);
	const float fl = floorf(u);
	const float f  = u - fl;
	int i0 = (int)fl, i1 = i0 + 1;
	i0 = i0 < 0 ? 0 : (i0 > p.pal_n - 1 ? p.pal_n - 1 : i0);
	i1 = i1 < 0 ? 0 : (i1 > p.pal_n - 1 ? p.pal_n - 1 : i1);
	const uint32_t a = pal[i0], b = pal[i1];
	uint32_t out = 0;
#pragma unroll
	for (int ch = 0; ch < 4; ch++) {
		const float c0 = (float)((a >> (8 * ch)) & 0xffu);
		const float c1 = (float)((b >> (8 * ch)) & 0xffu);
		const float c  = c0 + f * (c1 - c0);		/* -ffp-contract=off: mul, add */
		out |= ((uint32_t)(c + 0.5f) & 0xffu) << (8 * ch);
	}
	return out;
}

__global__ __launch_bounds__(256)
void k_colorize(const CmapParams p)
{
	extern __shared__ uint32_t pal[];
	for (int i = threadIdx.x; i < p.pal_n; i += 256)
		pal[i] = p.pal[i];
	__syncthreads();

	const int quads = p.n >> 2;				/* 4 pixels per thread */
	const int total = p.rows * quads;
	for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) {
		const int r = g / quads, c = (g - r * quads) << 2;
		const int sr = (p.row_base - r) & p.row_mask;
		const int sc = (c + (p.n >> 1)) & (p.n - 1);	/* fft-shift; stays 4-aligned and contiguous */
		const float4 t = *reinterpret_cast<const float4 *>(p.src + (size_t)sr * p.n + sc);
		uint4 o;
		o.x = lookup(t.x, p, pal); o.y = lookup(t.y, p, pal);
		o.z = lookup(t.z, p, pal); o.w = lookup(t.w, p, pal);
		*reinterpret_cast<uint4 *>(p.dst + (size_t)r * p.n + c) = o;
	}
}

} // namespace

extern "C" int fosphor_amd_cmap_generate(int which, uint32_t *rgba, int n)
{
	if (!rgba || n < 2)
		return -EINVAL;
	switch (which) {
	case FOSPHOR_AMD_CMAP_HISTOGRAM: palette_histogram(rgba, n); return 0;
	case FOSPHOR_AMD_CMAP_WATERFALL: palette_waterfall(rgba, n); return 0;
	case FOSPHOR_AMD_CMAP_PROG:      palette_prog(rgba, n);      return 0;
	}
	return -EINVAL;
}

extern "C" int fosphor_amd_colorize(struct fosphor *self, int image, const uint32_t *palette, int n,
                                    int use_defaults, float scale, float offset, int rows, uint32_t *d_rgba)
{
	struct fosphor_amd_buffers b;
	uint32_t own[256];
	uint32_t *d_pal = NULL;
	CmapParams p;
	hipStream_t st;

	if (!self || !d_rgba || (image != FOSPHOR_AMD_IMG_WATERFALL && image != FOSPHOR_AMD_IMG_HISTOGRAM))
		return -EINVAL;
	if (!palette) {
		n = 256;					/* gl.c:265-268 */
		fosphor_amd_cmap_generate(image == FOSPHOR_AMD_IMG_WATERFALL ? FOSPHOR_AMD_CMAP_WATERFALL
		                                                              : FOSPHOR_AMD_CMAP_HISTOGRAM, own, n);
		palette = own;
	}
	if (n < 2 || n > kPalMax)
		return -EINVAL;
	if (fosphor_amd_finish(self) < 0)			/* like fosphor_draw: wait for the compute side */
		return -EIO;
	if (fosphor_amd_get_buffers(self, &b))
		return -EIO;
	if (image == FOSPHOR_AMD_IMG_WATERFALL ? (rows < 1 || rows > b.wf_rows) : (rows != b.n_bins))
		return -EINVAL;
	if (use_defaults) {
		if (image == FOSPHOR_AMD_IMG_WATERFALL)
			fosphor_amd_priv_power(self, &scale, &offset);	/* gl.c:406-409 */
		else { scale = 1.1f; offset = 0.0f; }			/* gl.c:430-432 */
	}
	if (fosphor_amd_priv_palette(self, kPalMax, &d_pal) || !d_pal)
		return -EIO;
	st = (hipStream_t)fosphor_amd_stream(self);
	if (hipMemcpyAsync(d_pal, palette, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st) != hipSuccess)
		return -EIO;

	p.dst = d_rgba; p.pal = d_pal; p.pal_n = n;
	p.n = b.fft_len; p.rows = rows;
	p.scale = scale; p.offset = offset;
	if (image == FOSPHOR_AMD_IMG_WATERFALL) {
		p.src = b.d_waterfall;
		p.row_base = b.waterfall_pos - 1 + b.wf_rows;	/* kept non-negative before the mask */
		p.row_mask = b.wf_rows - 1;
	} else {
		p.src = b.d_histogram;
		p.row_base = b.n_bins - 1;
		p.row_mask = 0x7fffffff;
	}
	{
		const int total = rows * (p.n >> 2);
		int blocks = (total + 255) / 256;
		if (blocks > 4096) blocks = 4096;
		hipLaunchKernelGGL(k_colorize, dim3(blocks), dim3(256), sizeof(uint32_t) * n, st, p);
		if (hipGetLastError() != hipSuccess)
			return -EIO;
	}
	return hipStreamSynchronize(st) == hipSuccess ? 0 : -EIO;
}
