/*
 * fosphor_amd_cmap.h -- headless colour mapping of the plain device buffers
 *
 * The reference colours its two intensity textures in a GLSL fragment shader through a 256-entry
 * palette texture (lib/fosphor/gl_cmap.c, cmap_simple.glsl:41-47, palettes from gl_cmap_gen.c).
 * With the CL<->GL interop gone a front end (or a PNG dump for visual regression) gets the same
 * pictures without a GL context:
 *
 *   - the palette generators, on the host, bit-identical to the reference's tables;
 *   - an elementwise device pass   rgba = palette(( intensity + offset ) * scale)   that writes an
 *     RGBA8 image with one pixel per texel, fft-shifted (DC in the middle, gl.c:396-400) and with
 *     the newest waterfall row / the highest power bin on top (gl.c:403-404, 427-428).
 *
 * What a GL implementation leaves open -- the precision of the palette texture's GL_LINEAR
 * filter -- is defined here (float32, formula below); zooming / resampling to a window size is
 * the front end's business.
 */
#ifndef FOSPHOR_AMD_CMAP_H
#define FOSPHOR_AMD_CMAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct fosphor;

/* Palettes.  Entry i of n is the colour of intensity i / (n - 1), packed
 * (a << 24) | (b << 16) | (g << 8) | r as gl_cmap_gen.c:110-121 does (GL_RGBA bytes in memory). */
#define FOSPHOR_AMD_CMAP_HISTOGRAM 0	/* replaces fosphor_gl_cmap_histogram, gl_cmap_gen.c:150-178 */
#define FOSPHOR_AMD_CMAP_WATERFALL 1	/* replaces fosphor_gl_cmap_waterfall, gl_cmap_gen.c:181-198 */
#define FOSPHOR_AMD_CMAP_PROG      2	/* replaces fosphor_gl_cmap_prog,      gl_cmap_gen.c:271-322 */

/* Host only (no GPU needed).  n >= 2.  0, or -EINVAL. */
int fosphor_amd_cmap_generate(int which, uint32_t *rgba, int n);

/* Images */
#define FOSPHOR_AMD_IMG_WATERFALL 0	/* float[wf_rows][N] ring  -> rows x N pixels, newest row first */
#define FOSPHOR_AMD_IMG_HISTOGRAM 1	/* float[n_bins][N]        -> n_bins x N pixels, highest bin first */

/* Colour one of the instance's buffers into d_rgba (device memory, rows * N uint32, row-major).
 *
 *   pixel(r, c) = lookup( (texel + offset) * scale ),   texel column (c + N/2) mod N,
 *   texel row   (waterfall_pos - 1 - r) mod wf_rows   resp.   n_bins - 1 - r
 *   lookup(m):  u = m * n - 0.5;  i = floor(u), f = u - i;  entries clamp(i), clamp(i + 1) to
 *               [0, n - 1];  per channel  (uint8)(c0 + f * (c1 - c0) + 0.5)   in float32
 *               (GL_LINEAR + GL_CLAMP_TO_EDGE of gl_cmap.c:314-316); a NaN intensity takes entry 0.
 *
 * palette: n entries in HOST memory, or NULL for the reference's own 256-entry palette of that
 * image (gl.c:265-268).  scale/offset: pass use_defaults != 0 for the reference's values --
 * waterfall: the power range's (scale, offset) (gl.c:406-409, fosphor.c:131-152); histogram:
 * (1.1, 0) (gl.c:430-432).  rows: 1..wf_rows resp. must equal n_bins.
 * Synchronises with pending fosphor_process work first and returns when the image is complete.
 * 0, -EINVAL, -EIO. */
int fosphor_amd_colorize(struct fosphor *self, int image, const uint32_t *palette, int n,
                         int use_defaults, float scale, float offset, int rows, uint32_t *d_rgba);

#ifdef __cplusplus
}
#endif

#endif
