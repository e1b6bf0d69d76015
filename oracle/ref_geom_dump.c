/*
 * ref_geom_dump.c -- prints golden values of the reference's render-geometry helpers
 *
 * TEST INFRASTRUCTURE, build container only.  Linked (by `make -C oracle ref`) against the
 * reference's own lib/fosphor/fosphor.c compiled from where it lies; the functions exercised here
 * (fosphor_render_defaults / _refresh / _pos_inside, fosphor_pos2* / *2pos; fosphor.c:162-387) are
 * pure arithmetic on the two structs.  fosphor.c's other functions reference the OpenCL / GL halves
 * of the reference, which are not linked: the recipe leaves those symbols unresolved
 * (-Wl,--unresolved-symbols=ignore-all) and nothing here calls them.  The instance is a zeroed
 * struct fosphor (the reference's private.h) with the two range blocks filled in.
 *
 * Output: one JSON array on stdout, consumed by oracle/gen_golden.py -> tests/golden/render_geometry.json.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fosphor.h"
#include "private.h"

struct gcase { int db_ref, db_div; double center, span;
               int x, y, w, h, options; float ratio; int n_div; float fc, fs, ws; };

static const struct gcase cases[] = {
	{   0, 10,      0.0,     1.0,   0,  0, 1024, 1024, -1,    0.5f, 10, 0.5f,  1.0f,  1.0f },
	{ -20,  5,  100e6,     2e6,    0,  0, 1280,  720, -1,    0.5f, 10, 0.5f,  1.0f,  1.0f },
	{  10,  2,  433.92e6, 250e3,  40, 25,  800,  600, -1,    0.3f,  8, 0.25f, 0.5f,  0.25f },
	{   0, 10,  2.4e9,    20e6,    0,  0, 1920, 1080, FRO_LIVE | FRO_HISTO | FRO_LABEL_FREQ | FRO_LABEL_PWR, 0.5f, 10, 0.5f, 1.0f, 1.0f },
	{   0, 10,  2.4e9,    20e6,    0,  0, 1920, 1080, FRO_WATERFALL | FRO_LABEL_TIME | FRO_COLOR_SCALE, 0.5f, 12, 0.7f, 0.1f, 0.5f },
	{ -30,  3,  10.7e6,   48e3,   10, 10,  640,  480, FRO_LIVE | FRO_MAX_HOLD | FRO_HISTO | FRO_WATERFALL, 0.8f, 6, 0.5f, 1.0f, 1.0f },
	{   0, 10,  -5e6,     1e6,     0,  0,  333,  777, -1,    0.41f, 7, 0.123f, 0.2f, 0.9f },
	{   0, 10,   1e3,     100.0,   0,  0,  200,  150, -1,    0.5f,  4, 0.5f,  1.0f,  1.0f },
};

int main(void)
{
	const int n = (int)(sizeof(cases) / sizeof(cases[0]));
	printf("[\n");
	for (int k = 0; k < n; k++) {
		const struct gcase *c = &cases[k];
		struct fosphor *self = calloc(1, sizeof(*self));
		struct fosphor_render r;
		self->power.db_ref = c->db_ref; self->power.db_per_div = c->db_div;
		self->frequency.center = c->center; self->frequency.span = c->span;
		memset(&r, 0, sizeof(r));
		fosphor_render_defaults(&r);
		printf(" {\"db_ref\": %d, \"db_per_div\": %d, \"center\": %.17g, \"span\": %.17g,\n", c->db_ref, c->db_div, c->center, c->span);
		printf("  \"defaults\": [%d, %d, %d, %d, %d, %.9g, %d, %.9g, %.9g, %.9g],\n", r.pos_x, r.pos_y, r.width, r.height,
		       r.options, r.histo_wf_ratio, r.freq_n_div, r.freq_center, r.freq_span, r.wf_span);
		r.pos_x = c->x; r.pos_y = c->y; r.width = c->w; r.height = c->h;
		if (c->options >= 0) r.options = c->options;
		r.histo_wf_ratio = c->ratio; r.freq_n_div = c->n_div; r.freq_center = c->fc; r.freq_span = c->fs; r.wf_span = c->ws;
		r._wf_pos = 300;
		printf("  \"user\": [%d, %d, %d, %d, %d, %.9g, %d, %.9g, %.9g, %.9g],\n", r.pos_x, r.pos_y, r.width, r.height,
		       r.options, r.histo_wf_ratio, r.freq_n_div, r.freq_center, r.freq_span, r.wf_span);
		fosphor_render_refresh(&r);
		printf("  \"private\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g],\n", r._x_div, r._x[0], r._x[1], r._x_label,
		       r._y_histo_div, r._y_histo[0], r._y_histo[1], r._y_wf[0], r._y_wf[1], r._y_label);
		printf("  \"probes\": [");
		for (int i = 0; i < 9; i++) {
			const int x = c->x + (c->w * i) / 8 - (i == 0 ? 3 : 0) + (i == 8 ? 3 : 0);
			const int y = c->y + (c->h * i) / 8;
			const double f = fosphor_pos2freq(self, &r, x);
			const float pw = fosphor_pos2pwr(self, &r, y);
			const int sm = fosphor_pos2samp(self, &r, y);
			printf("%s[%d, %d, %.17g, %d, %.9g, %d, %d, %d, %d]", i ? ", " : "", x, y, f, fosphor_freq2pos(self, &r, f),
			       pw, fosphor_pwr2pos(self, &r, pw), sm, fosphor_samp2pos(self, &r, sm), fosphor_render_pos_inside(&r, x, y));
		}
		printf("]}%s\n", k + 1 < n ? "," : "");
		free(self);
	}
	printf("]\n");
	return 0;
}
