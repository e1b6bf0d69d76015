/*
 * fosphor_amd_axis.h -- axis labels for a headless front end
 *
 * The reference formats its frequency-axis labels in lib/fosphor/axis.c (freq_axis_build /
 * freq_axis_render, :63-162) and prints the power labels inline (gl.c:600-606).  A front end
 * that draws the plain buffers of fosphor_amd_get_buffers() itself gets the same strings here.
 * Host only; no GPU involved.
 */
#ifndef FOSPHOR_AMD_AXIS_H
#define FOSPHOR_AMD_AXIS_H

#ifdef __cplusplus
extern "C" {
#endif

struct fosphor;
struct fosphor_render;

/* Same fields as the reference's struct freq_axis (axis.h:20-30) */
struct fosphor_amd_freq_axis
{
	double center;
	double span;
	double step;
	int    mode;		/* 0 count, 1 relative, 2 absolute (axis.c:27-29) */
	char   abs_fmt[16];
	double abs_scale;
	char   rel_fmt[16];
	double rel_step;
};

/* replaces freq_axis_build, axis.c:63-136 */
void fosphor_amd_freq_axis_build(struct fosphor_amd_freq_axis *fx, double center, double span, int n_div);
/* replaces freq_axis_render, axis.c:138-162: label of division `step` (0 = centre) into str (>= 32 bytes) */
void fosphor_amd_freq_axis_render(const struct fosphor_amd_freq_axis *fx, char *str, int step);

/* The freq_n_div + 1 labels along the frequency axis of `render`, left to right, for the
 * instance's frequency range and the render's zoom (gl.c:554-575, 640-643).  Returns the number
 * of labels written (<= max_labels), or -EINVAL. */
int fosphor_amd_freq_labels(struct fosphor *self, const struct fosphor_render *render,
                            char (*labels)[32], int max_labels);

/* The 11 power-axis values in dB, bottom to top (gl.c:600-606). */
int fosphor_amd_power_labels(struct fosphor *self, int db[11]);

#ifdef __cplusplus
}
#endif

#endif
