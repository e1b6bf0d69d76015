/*
 * fosphor.h -- drop-in C API of the MI355X-native fosphor compute core
 *
 * Source-compatible with the reference's public header
 * (lib/fosphor/fosphor.h:26-105): same function names, same argument meaning,
 * same struct layouts, same error conventions -- so base_sink_c_impl.cc:27-30
 * (`extern "C" { #include "fosphor/fosphor.h" }`) and lib/fosphor/main.c keep
 * compiling against libfosphor_amd.so unchanged.
 *
 * What differs behind the API: the OpenCL compute path
 * (lib/fosphor/{fft.cl,display.cl,cl.c}) is replaced by HIP kernels for
 * gfx950, and the CL<->GL interop is severed: results live in plain device
 * buffers (see fosphor_amd.h, fosphor_amd_get_buffers) that any front end can
 * map or copy.  fosphor_draw() therefore does not draw; it is the
 * synchronisation point it always was (fosphor.c:98-105 -> cl.c:970-1061).
 *
 * Each declaration cites the reference interface it replaces.
 */
#ifndef FOSPHOR_AMD_FOSPHOR_H
#define FOSPHOR_AMD_FOSPHOR_H

#ifdef __cplusplus
extern "C" {
#endif

struct fosphor;			/* opaque; reference: private.h:30-55 */
struct fosphor_render;

/* ---- life cycle -------------------------------------------------------- */

/* Replaces fosphor_init (fosphor.h:26, fosphor.c:29-74).  Reference geometry:
 * 1024-point FFT, 128 histogram bins, 1024 waterfall rows; default Hamming
 * window; power range (0 dB ref, 10 dB/div).  Returns NULL on any failure
 * after printing the cause to stderr (fosphor.c:70-73, cl.c:838-842) --
 * including "no HIP device": there is no CPU fallback. */
struct fosphor *fosphor_init(void);

/* Replaces fosphor_release (fosphor.h:27, fosphor.c:76-90).  NULL is allowed. */
void fosphor_release(struct fosphor *self);

/* ---- data path --------------------------------------------------------- */

/* Replaces fosphor_process (fosphor.h:29, fosphor.c:92-96 -> cl.c:870-968).
 * samples: `len` interleaved fp32 (re, im) pairs in HOST memory (gr_complex).
 * len counts complex samples; it must be a multiple of 16*1024 and at most
 * 1024*1024, otherwise -EINVAL (cl.c:882-886).  Device error: -EIO.
 * Asynchronous: work is queued on the instance's HIP stream.  The caller may
 * reuse `samples` as soon as the call returns (base_sink_c_impl.cc:168-174):
 * the data has been copied into a pinned staging ring by then. */
int fosphor_process(struct fosphor *self, void *samples, int len);

/* Replaces fosphor_draw (fosphor.h:30, fosphor.c:98-105).  Waits for every
 * queued fosphor_process, like fosphor_cl_finish's clFinish (cl.c:1052), and
 * stores the waterfall ring position into render->_wf_pos (fosphor.c:103).
 * Nothing is drawn: the renderer is out of scope, the buffers are exported. */
void fosphor_draw(struct fosphor *self, struct fosphor_render *render);

/* ---- settings ---------------------------------------------------------- */

/* Replace fosphor_set_fft_window_default / fosphor_set_fft_window
 * (fosphor.h:32-33, fosphor.c:108-128).  win: 1024 floats, copied; uploaded
 * lazily before the next process (cl.c:889-900, 1064-1071). */
void fosphor_set_fft_window_default(struct fosphor *self);
void fosphor_set_fft_window(struct fosphor *self, float *win);

/* Replaces fosphor_set_power_range (fosphor.h:35, fosphor.c:131-152 ->
 * cl.c:1081-1089): db0 = db_ref - 10*db_per_div; offset = -(log10(N) + db0/20);
 * scale = 20/(db_ref - db0); histogram scale = scale * n_bins. */
void fosphor_set_power_range(struct fosphor *self, int db_ref, int db_per_div);

/* Replaces fosphor_set_frequency_range (fosphor.h:36-37, fosphor.c:154-160).
 * Pure bookkeeping for the pixel<->frequency mapping. */
void fosphor_set_frequency_range(struct fosphor *self, double center, double span);

/* ---- render geometry (layout kept; fosphor.h:42-90) --------------------- */

#define FOSPHOR_MAX_CHANNELS	8

struct fosphor_channel
{
	int   enabled;
	float center;
	float width;
};

#define FRO_LIVE	(1<<0)
#define FRO_MAX_HOLD	(1<<1)
#define FRO_HISTO	(1<<2)
#define FRO_WATERFALL	(1<<3)
#define FRO_LABEL_FREQ	(1<<4)
#define FRO_LABEL_PWR	(1<<5)
#define FRO_LABEL_TIME	(1<<6)
#define FRO_CHANNELS	(1<<7)
#define FRO_COLOR_SCALE	(1<<8)

struct fosphor_render
{
	/* user fields */
	int   pos_x;
	int   pos_y;
	int   width;
	int   height;
	int   options;
	float histo_wf_ratio;
	int   freq_n_div;
	float freq_center;
	float freq_span;
	float wf_span;

	struct fosphor_channel channels[FOSPHOR_MAX_CHANNELS];

	/* private fields */
	int   _wf_pos;

	float _x_div;
	float _x[2];
	float _x_label;

	float _y_histo_div;
	float _y_histo[2];
	float _y_wf[2];
	float _y_label;
};

/* Replace fosphor_render_defaults / fosphor_render_refresh
 * (fosphor.h:92-93, fosphor.c:162-272). */
void fosphor_render_defaults(struct fosphor_render *render);
void fosphor_render_refresh(struct fosphor_render *render);

/* Replace the position-mapping helpers (fosphor.h:98-105, fosphor.c:275-387). */
double fosphor_pos2freq(struct fosphor *self, struct fosphor_render *render, int x);
float  fosphor_pos2pwr (struct fosphor *self, struct fosphor_render *render, int y);
int    fosphor_pos2samp(struct fosphor *self, struct fosphor_render *render, int y);
int    fosphor_freq2pos(struct fosphor *self, struct fosphor_render *render, double freq);
int    fosphor_pwr2pos (struct fosphor *self, struct fosphor_render *render, float pwr);
int    fosphor_samp2pos(struct fosphor *self, struct fosphor_render *render, int time);
int    fosphor_render_pos_inside(struct fosphor_render *render, int x, int y);

#ifdef __cplusplus
}
#endif

#endif /* FOSPHOR_AMD_FOSPHOR_H */
