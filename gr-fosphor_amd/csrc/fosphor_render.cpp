/*
 * fosphor_render.cpp -- render-geometry and pixel<->unit helpers of the drop-in API
 *
 * Pure CPU arithmetic kept so that front ends written against the reference's
 * fosphor.h keep linking (lib/fosphor/fosphor.c:162-387).  No drawing happens in this
 * library; these only lay out where a front end would place the histogram / waterfall
 * panes and translate pixel positions into frequency / power / sample units.
 */
#include <math.h>

#include "../../include/fosphor.h"

/* The two settings blocks these helpers read; filled by fosphor_set_power_range /
 * fosphor_set_frequency_range in fosphor_api.cpp through the accessors below. */
extern "C" void fosphor_amd_priv_ranges(struct fosphor *self, int *db_ref, int *db_per_div,
                                        double *center, double *span);

namespace {

constexpr int   kMargin      = 10;	/* outer margin, px */
constexpr int   kLabelMargin = 30;	/* extra left margin when power/time labels are shown */
constexpr int   kScaleMargin = 10;	/* extra right margin for the colour scale */
constexpr int   kDivPx       = 80;	/* minimum width of one frequency division */
constexpr float kWfSamples   = 1024.0f * 1024.0f;	/* FFT_LEN * waterfall rows */

inline bool shows_spectrum(const fosphor_render *r) { return (r->options & (FRO_LIVE | FRO_MAX_HOLD | FRO_HISTO)) != 0; }
inline bool shows_waterfall(const fosphor_render *r) { return (r->options & FRO_WATERFALL) != 0; }

struct view { double center, span; };

view freq_view(struct fosphor *self, const fosphor_render *r)
{
	int a, b; double c, s;
	fosphor_amd_priv_ranges(self, &a, &b, &c, &s);
	view v;
	v.center = c + s * (double)(r->freq_center - 0.5f);
	v.span   = s * (double)r->freq_span;
	return v;
}

} // namespace

/* fosphor.c:162-184 */
extern "C" void fosphor_render_defaults(struct fosphor_render *r)
{
	r->pos_x = 0;
	r->pos_y = 0;
	r->width = 1024;
	r->height = 1024;
	r->options = FRO_LIVE | FRO_MAX_HOLD | FRO_HISTO | FRO_WATERFALL |
	             FRO_LABEL_FREQ | FRO_LABEL_PWR | FRO_LABEL_TIME | FRO_COLOR_SCALE;
	r->histo_wf_ratio = 0.5f;
	r->freq_n_div = 10;
	r->freq_center = 0.5f;
	r->freq_span = 1.0f;
	r->wf_span = 1.0f;
}

/* fosphor.c:186-272 */
extern "C" void fosphor_render_refresh(struct fosphor_render *r)
{
	const bool spec = shows_spectrum(r), wf = shows_waterfall(r);

	/* horizontal split */
	const int left  = kMargin + ((r->options & (FRO_LABEL_PWR | FRO_LABEL_TIME)) ? kLabelMargin : 0);
	const int right = kMargin + ((r->options & FRO_COLOR_SCALE) ? kScaleMargin : 0);
	int usable = r->width - (left + right);

	int ndiv = (usable / kDivPx) & ~1;
	if (ndiv > 10) ndiv = 10;
	if (ndiv < 2)  ndiv = 2;
	r->freq_n_div = ndiv;

	int div_px = usable / ndiv;
	int slack  = usable - ndiv * div_px;

	r->_x_div   = (float)div_px;
	r->_x[0]    = r->pos_x + (float)left + (float)(slack / 2);
	r->_x[1]    = r->_x[0] + (ndiv * r->_x_div) + 1.0f;
	r->_x_label = r->_x[0] - 5.0f;

	/* vertical split */
	float top = r->pos_y + (float)r->height - 10.0f;
	float bot = r->pos_y + 10.0f;

	if (spec) {
		int reserved = 20;			/* frame + spectrum spacing */
		if (wf) reserved += 10;
		if (r->options & FRO_LABEL_FREQ) reserved += 10;

		if (wf) {
			usable = (int)((float)(r->height - reserved) * r->histo_wf_ratio);
			div_px = usable / 10;
			slack  = 0;
		} else {
			usable = r->height - reserved;
			div_px = usable / 10;
			slack  = usable - 10 * div_px;
		}
		r->_y_histo_div = (float)div_px;
		r->_y_histo[1]  = top - (float)(slack / 2);
		r->_y_histo[0]  = r->_y_histo[1] - (10.0f * r->_y_histo_div) - 1.0f;
		top = r->_y_histo[0] - (float)(slack / 2) - 10.0f;
	} else {
		r->_y_histo_div = 0.0f;
		r->_y_histo[0] = r->_y_histo[1] = 0.0f;
	}

	if (r->options & FRO_LABEL_FREQ) {
		if (r->options & FRO_HISTO) { r->_y_label = top; top -= 10.0f; }
		else                        { r->_y_label = bot; bot += 10.0f; }
	} else {
		r->_y_label = 0.0f;
	}

	r->_y_wf[1] = wf ? top : 0.0f;
	r->_y_wf[0] = wf ? bot : 0.0f;
}

/* Pixel centres are at integer + 0.5; the frequency axis spans
 * [center - span/2, center + span/2] between _x[0] and _x[1] (fosphor.c:275-302). */
extern "C" double fosphor_pos2freq(struct fosphor *self, struct fosphor_render *r, int x)
{
	const float rel = (((float)x + 0.5f) - r->_x[0]) / (r->_x[1] - r->_x[0]);
	const view v = freq_view(self, r);
	return v.center + v.span * (double)(rel - 0.5f);
}

/* fosphor.c:304-313 */
extern "C" float fosphor_pos2pwr(struct fosphor *self, struct fosphor_render *r, int y)
{
	int db_ref, db_div; double c, s;
	fosphor_amd_priv_ranges(self, &db_ref, &db_div, &c, &s);
	const float extent = r->_y_histo[1] - r->_y_histo[0] - 1.0f;
	const float rel = ((float)y - r->_y_histo[0]) / extent;
	return db_ref - 10.0f * db_div * (1.0f - rel);
}

/* fosphor.c:315-324 */
extern "C" int fosphor_pos2samp(struct fosphor *self, struct fosphor_render *r, int y)
{
	(void)self;
	const float extent = r->_y_wf[1] - r->_y_wf[0] - 1.0f;
	const float rel = ((float)y - r->_y_wf[0]) / extent;
	return (int)((1.0f - rel) * kWfSamples) * r->wf_span;
}

/* fosphor.c:326-337 */
extern "C" int fosphor_freq2pos(struct fosphor *self, struct fosphor_render *r, double freq)
{
	const view v = freq_view(self, r);
	const double rel = (freq - v.center) / v.span;
	const float extent = r->_x[1] - r->_x[0];
	return (int)roundf(r->_x[0] + (float)(rel + 0.5) * extent - 0.5f);
}

/* fosphor.c:339-346 */
extern "C" int fosphor_pwr2pos(struct fosphor *self, struct fosphor_render *r, float pwr)
{
	int db_ref, db_div; double c, s;
	fosphor_amd_priv_ranges(self, &db_ref, &db_div, &c, &s);
	const float rel = (db_ref - pwr) / (10.0f * db_div);
	const float extent = r->_y_histo[1] - r->_y_histo[0] - 1.0f;
	return (int)roundf(r->_y_histo[0] + (1.0f - rel) * extent);
}

/* fosphor.c:348-356 */
extern "C" int fosphor_samp2pos(struct fosphor *self, struct fosphor_render *r, int time)
{
	(void)self;
	const float rel = (float)time / (kWfSamples * r->wf_span);
	const float extent = r->_y_wf[1] - r->_y_wf[0] - 1.0f;
	return (int)roundf(r->_y_wf[0] + (1.0f - rel) * extent);
}

/* bit 0: inside the x range; bit 1: inside the spectrum pane; bit 2: inside the waterfall
 * pane (fosphor.c:358-387) */
extern "C" int fosphor_render_pos_inside(struct fosphor_render *r, int x, int y)
{
	const float fx = (float)x, fy = (float)y;
	int mask = 0;
	if (fx >= r->_x[0] && fx < r->_x[1])
		mask |= 1;
	if (shows_spectrum(r) && fy >= r->_y_histo[0] && fy < r->_y_histo[1])
		mask |= 2;
	if (shows_waterfall(r) && fy >= r->_y_wf[0] && fy < r->_y_wf[1])
		mask |= 4;
	return mask;
}

/* ------------------------------------------------------------------------ */
/* Axis labels (include/fosphor_amd_axis.h)                                   */
/* ------------------------------------------------------------------------ */

#include <errno.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fosphor_amd_axis.h"

namespace {

/* Power of 1000 by which a value is divided so that at most `digits` digits stay in front of the
 * decimal point, as an exponent of ten (0, 3, 6, ...).  axis.c:32-47: the magnitude is taken
 * with the FLOAT log10 of the value and truncated towards zero. */
int si_exponent(double val, int digits)
{
	const int mag = (int)log10f((float)fabs(val));
	return (mag >= digits) ? 3 * ((mag - digits + 3) / 3) : 0;
}

const char *si_prefix(int exp10)
{
	static const char *const names[5] = { "", "k", "M", "G", "T" };	/* ' ' is dropped by the callers, axis.c:99-100 */
	return names[exp10 / 3];
}

/* decimals needed to print val exactly, up to 21; -1 if none does (axis.c:49-61) */
int decimals_needed(double val)
{
	for (int c = 0; c < 22; c++) {
		const double v = val * pow(10, c);
		if (fabs(v - round(v)) == 0.0)
			return c;
	}
	return -1;
}

} // namespace

extern "C" void fosphor_amd_freq_axis_build(struct fosphor_amd_freq_axis *fx, double center, double span, int n_div)
{
	fx->center = center;
	fx->span   = span;
	fx->step   = span / n_div;

	/* count / relative / absolute, axis.c:71-80 (float log10 of the doubles, as there) */
	if (span == 0.0)
		fx->mode = 0;
	else if (center == 0.0)
		fx->mode = 1;
	else
		fx->mode = (floor(log10f((float)fx->step)) < floor(log10f((float)fx->center)) - 4) ? 1 : 2;

	if (center != 0.0) {					/* axis.c:83-113 */
		const double hi = fabs(center + span / 2.0), lo = fabs(center - span / 2.0);
		const double big = hi > lo ? hi : lo;
		const int e = si_exponent(big, 4);
		unsigned int x, y, z;

		fx->abs_scale = 1.0 / powf(10.0, e);
		x = (int)floor(log10f((float)big)) - e + 1;
		y = decimals_needed(fx->center * fx->abs_scale);
		z = decimals_needed(fx->step * fx->abs_scale);
		if (z > y) y = z;
		if (x + y > 6) y = 6 - x;
		snprintf(fx->abs_fmt, sizeof(fx->abs_fmt), "%%.%dlf%s", y, si_prefix(e));
	}

	if (fx->mode == 1) {					/* axis.c:116-135 */
		const double max_dev = fx->step * 5;
		const int e = si_exponent(max_dev, 3);
		unsigned int x, y;

		fx->rel_step = fx->step / powf(10.0, e);
		x = (int)floor(log10f((float)max_dev)) - e + 1;
		y = decimals_needed(fx->rel_step);
		if (x + y > 4) y = 4 - x;
		snprintf(fx->rel_fmt, sizeof(fx->rel_fmt), "%%+.%dlf%s", y, si_prefix(e));
	}
}

extern "C" void fosphor_amd_freq_axis_render(const struct fosphor_amd_freq_axis *fx, char *str, int step)
{
	if (step && fx->mode == 0)				/* axis.c:141-145 */
		snprintf(str, 32, "%+d", step);
	else if (!step && fx->center == 0.0)			/* :147-151 */
		snprintf(str, 32, "0");
	else if (step && fx->mode == 1)				/* :153-157 */
		snprintf(str, 32, fx->rel_fmt, step * fx->rel_step);
	else							/* :159-160 */
		snprintf(str, 32, fx->abs_fmt, (fx->center + step * fx->step) * fx->abs_scale);
}

extern "C" int fosphor_amd_freq_labels(struct fosphor *self, const struct fosphor_render *render,
                                       char (*labels)[32], int max_labels)
{
	struct fosphor_amd_freq_axis fx;
	int a, b; double c, s;
	if (!self || !render || !labels || render->freq_n_div < 1 || max_labels < render->freq_n_div + 1)
		return -EINVAL;
	fosphor_amd_priv_ranges(self, &a, &b, &c, &s);
	if (render->freq_center != 0.5f || render->freq_span != 1.0f) {	/* gl.c:555-565 */
		const view v = freq_view(self, render);
		fosphor_amd_freq_axis_build(&fx, v.center, v.span, render->freq_n_div);
	} else {								/* gl.c:566-575: the numbers as given */
		fosphor_amd_freq_axis_build(&fx, c, s, render->freq_n_div);
	}
	for (int i = 0; i <= render->freq_n_div; i++)
		fosphor_amd_freq_axis_render(&fx, labels[i], i - render->freq_n_div / 2);	/* gl.c:640-643 */
	return render->freq_n_div + 1;
}

extern "C" int fosphor_amd_power_labels(struct fosphor *self, int db[11])
{
	int db_ref, db_div; double c, s;
	if (!self || !db)
		return -EINVAL;
	fosphor_amd_priv_ranges(self, &db_ref, &db_div, &c, &s);
	for (int i = 0; i < 11; i++)
		db[i] = db_ref - (10 - i) * db_div;			/* gl.c:604 */
	return 0;
}
