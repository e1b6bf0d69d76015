/*
 * ref_shim.cpp -- run the reference's OpenCL kernels on the host CPU
 *
 * TEST INFRASTRUCTURE.  Built only where /root/reference exists (this
 * container), by oracle/Makefile, into oracle/_ref/libfosphor_ref.so:
 *
 *   clang -x cl ... -target x86_64 -c /root/reference/lib/fosphor/fft.cl
 *   clang -x cl ... -target x86_64 -c /root/reference/lib/fosphor/display.cl
 *   clang++ ref_shim.cpp fft.o display.o -shared -o _ref/libfosphor_ref.so
 *
 * The kernel bodies are the reference's, compiled from where they lie; no
 * reference source is copied.  What this file supplies is what an OpenCL
 * runtime would: work-item ids, barrier(), images, and the math built-ins.
 * No OpenCL CPU runtime exists in the image, so those are ours -- and the
 * OpenCL spec leaves native_sin/native_cos/native_powr/native_recip
 * precision implementation-defined and hypot/log10 at <=4/<=3 ulp, i.e. the
 * reference does not pin them either.  Two bindings, switchable at run time:
 *
 *   binding 1 "portable": include/fosphor_portable_math.h  (the pinned oracle)
 *   binding 0 "glibc":    sinf cosf hypotf log10f roundf   (informational:
 *                         what a typical CPU OpenCL runtime would produce)
 *
 * powr -> powf and recip -> 1/x in both bindings.
 *
 * Work-items run as ucontext fibers, one work-group at a time (the kernels'
 * __local arrays are single static objects), switched at barrier().
 *
 * Host-side state handling follows lib/fosphor/cl.c:406-465 (first-run
 * fills), :870-968 (process), :1081-1089 (histogram range) and
 * lib/fosphor/fosphor.c:108-152 (default window, power range).
 */
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ucontext.h>
#include <vector>

#include "../include/fosphor_portable_math.h"

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int   int2   __attribute__((ext_vector_type(2)));

/* ---- kernel entry points (plain C symbols in fft.o / display.o) --------- */
struct Img { float *d; int w, h; };

extern "C" void fft1D_1024(const float2 *input, float2 *output, const float *win);
extern "C" void fft1D_512(const float2 *input, float2 *output, const float *win);
extern "C" void display(const float2 *fft, unsigned log2len, unsigned batch,
                        Img *wf, unsigned wf_offset, Img *hist_r, Img *hist_w,
                        float t0r, float t0d, float histo_scale, float histo_ofs,
                        float2 *spectrum, float live_alpha);

/* ---- fiber scheduler ----------------------------------------------------- */
namespace {

int g_binding = 1;

struct WI { size_t lid[3], gid[3]; };
size_t g_lsize[3], g_gsize[3];
WI *g_cur;

constexpr size_t STACK = 64 * 1024;
ucontext_t g_sched;
std::vector<ucontext_t> g_ctx;
std::vector<char> g_stacks;
std::vector<char> g_done;
void (*g_body)();

void trampoline()
{
	g_body();
}

/* Run `count` work-items of one work-group to completion, round-robin
 * between barriers. */
void run_group(std::vector<WI> &items, void (*body)())
{
	const size_t count = items.size();
	g_body = body;
	if (g_ctx.size() < count) {
		g_ctx.resize(count);
		g_stacks.resize(count * STACK);
	}
	g_done.assign(count, 0);
	for (size_t i = 0; i < count; i++) {
		getcontext(&g_ctx[i]);
		g_ctx[i].uc_stack.ss_sp = &g_stacks[i * STACK];
		g_ctx[i].uc_stack.ss_size = STACK;
		g_ctx[i].uc_link = &g_sched;
		makecontext(&g_ctx[i], trampoline, 0);
	}
	size_t remaining = count;
	std::vector<char> finished(count, 0);
	while (remaining) {
		for (size_t i = 0; i < count; i++) {
			if (finished[i]) continue;
			g_cur = &items[i];
			g_done[i] = 1;			/* cleared by barrier() if it yields */
			swapcontext(&g_sched, &g_ctx[i]);
			if (g_done[i]) { finished[i] = 1; remaining--; }
		}
	}
}

} // namespace

/* ---- OpenCL built-ins the kernels import (Itanium-mangled) -------------- */
#define CLSYM(name) __asm__(name)

size_t cl_get_local_id(unsigned d)    CLSYM("_Z12get_local_idj");
size_t cl_get_global_id(unsigned d)   CLSYM("_Z13get_global_idj");
size_t cl_get_local_size(unsigned d)  CLSYM("_Z14get_local_sizej");
size_t cl_get_global_size(unsigned d) CLSYM("_Z15get_global_sizej");
void   cl_barrier(unsigned flags)     CLSYM("_Z7barrierj");
float  cl_native_sin(float x)         CLSYM("_Z10native_sinf");
float  cl_native_cos(float x)         CLSYM("_Z10native_cosf");
float  cl_native_powr(float x, float y) CLSYM("_Z11native_powrff");
float  cl_native_recip(float x)       CLSYM("_Z12native_recipf");
float  cl_max(float x, float y)       CLSYM("_Z3maxff");
float  cl_clamp(float x, float lo, float hi) CLSYM("_Z5clampfff");
float  cl_hypot(float x, float y)     CLSYM("_Z5hypotff");
float  cl_log10(float x)              CLSYM("_Z5log10f");
float  cl_round(float x)              CLSYM("_Z5roundf");
int    cl_isfinite(float x)           CLSYM("_Z8isfinitef");
unsigned cl_atomic_inc(volatile unsigned *p) CLSYM("_Z10atomic_incPU7CLlocalVj");
float4 cl_read_imagef(Img *img, void *sampler, int2 c) CLSYM("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
void   cl_write_imagef(Img *img, int2 c, float4 v) CLSYM("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
int    cl_get_image_height(Img *img)  CLSYM("_Z16get_image_height14ocl_image2d_wo");
extern "C" void *__translate_sampler_initializer(int);

size_t cl_get_local_id(unsigned d)    { return g_cur->lid[d]; }
size_t cl_get_global_id(unsigned d)   { return g_cur->gid[d]; }
size_t cl_get_local_size(unsigned d)  { return g_lsize[d]; }
size_t cl_get_global_size(unsigned d) { return g_gsize[d]; }

void cl_barrier(unsigned)
{
	/* find which fiber we are: g_cur points into the items vector */
	WI *me = g_cur;
	extern std::vector<WI> *g_items_ptr;
	size_t idx = (size_t)(me - g_items_ptr->data());
	g_done[idx] = 0;
	swapcontext(&g_ctx[idx], &g_sched);
	g_cur = me;
}
std::vector<WI> *g_items_ptr;

float cl_native_sin(float x) { return g_binding ? fpm_sinf(x) : sinf(x); }
float cl_native_cos(float x) { return g_binding ? fpm_cosf(x) : cosf(x); }
float cl_native_powr(float x, float y) { return powf(x, y); }
float cl_native_recip(float x) { return 1.0f / x; }
float cl_max(float x, float y) { return (x < y) ? y : x; }
float cl_clamp(float x, float lo, float hi) { float t = (x < lo) ? lo : x; return (hi < t) ? hi : t; }
float cl_hypot(float x, float y) { return g_binding ? fpm_hypotf(x, y) : hypotf(x, y); }
float cl_log10(float x) { return g_binding ? fpm_log10f(x) : log10f(x); }
/* (int)round(v) of a non-finite v is undefined in OpenCL C; x86 cvttss2si
 * yields INT_MIN, which display.cl:163-165 clamps to bin 0.  Returning a
 * large negative finite value makes that explicit and compiler-independent. */
float cl_round(float x)
{
	if (!(fabsf(x) <= 3.0e38f)) return -1.0e9f;
	return g_binding ? fpm_roundf(x) : roundf(x);
}
int   cl_isfinite(float x) { return std::isfinite(x) ? 1 : 0; }
unsigned cl_atomic_inc(volatile unsigned *p) { unsigned o = *p; *p = o + 1; return o; }

float4 cl_read_imagef(Img *img, void *, int2 c)
{
	int x = c.x < 0 ? 0 : (c.x >= img->w ? img->w - 1 : c.x);	/* CLK_ADDRESS_CLAMP_TO_EDGE */
	int y = c.y < 0 ? 0 : (c.y >= img->h ? img->h - 1 : c.y);
	float4 r = { img->d[(size_t)y * img->w + x], 0.0f, 0.0f, 1.0f };
	return r;
}
void cl_write_imagef(Img *img, int2 c, float4 v)
{
	if (c.x < 0 || c.y < 0 || c.x >= img->w || c.y >= img->h) return;
	img->d[(size_t)c.y * img->w + c.x] = v.x;
}
int cl_get_image_height(Img *img) { return img->h; }
extern "C" void *__translate_sampler_initializer(int) { static int dummy; return &dummy; }

/* ---- kernel launch emulation --------------------------------------------- */
namespace {

const float2 *k_fft_in; float2 *k_fft_out; const float *k_fft_win; int k_fft_n;
void body_fft()
{
	if (k_fft_n == 1024) fft1D_1024(k_fft_in, k_fft_out, k_fft_win);
	else                 fft1D_512(k_fft_in, k_fft_out, k_fft_win);
}

struct DispArgs {
	const float2 *fft; unsigned log2len, batch; Img *wf; unsigned wf_offset; Img *hist;
	float t0r, t0d, hs, ho; float2 *spec; float alpha;
} k_disp;
void body_display()
{
	display(k_disp.fft, k_disp.log2len, k_disp.batch, k_disp.wf, k_disp.wf_offset,
	        k_disp.hist, k_disp.hist, k_disp.t0r, k_disp.t0d, k_disp.hs, k_disp.ho,
	        k_disp.spec, k_disp.alpha);
}

/* cl.c:913-919: global (N/8, n_spectra), local (N/8, 1) */
void launch_fft(int n, const float *in, float *out, const float *win, int n_spectra)
{
	const int wg = n / 8;
	std::vector<WI> items(wg);
	g_items_ptr = &items;
	g_lsize[0] = wg; g_lsize[1] = 1; g_lsize[2] = 1;
	g_gsize[0] = wg; g_gsize[1] = n_spectra; g_gsize[2] = 1;
	k_fft_in = (const float2 *)in; k_fft_out = (float2 *)out; k_fft_win = win; k_fft_n = n;
	for (int s = 0; s < n_spectra; s++) {
		for (int l = 0; l < wg; l++) {
			items[l].lid[0] = l; items[l].lid[1] = 0; items[l].lid[2] = 0;
			items[l].gid[0] = l; items[l].gid[1] = s; items[l].gid[2] = 0;
		}
		run_group(items, body_fft);
	}
}

/* cl.c:945-950: global (N, 16), local (16, 16) */
void launch_display(int n)
{
	std::vector<WI> items(256);
	g_items_ptr = &items;
	g_lsize[0] = 16; g_lsize[1] = 16; g_lsize[2] = 1;
	g_gsize[0] = n;  g_gsize[1] = 16; g_gsize[2] = 1;
	for (int g = 0; g < n / 16; g++) {
		for (int l1 = 0; l1 < 16; l1++)
			for (int l0 = 0; l0 < 16; l0++) {
				WI &w = items[l1 * 16 + l0];
				w.lid[0] = l0; w.lid[1] = l1; w.lid[2] = 0;
				w.gid[0] = 16 * g + l0; w.gid[1] = l1; w.gid[2] = 0;
			}
		run_group(items, body_display);
	}
}

} // namespace

/* ---- exported C API ------------------------------------------------------- */
struct ref_state {
	float win[1024];
	std::vector<float> wf, hist, spectrum, fft_out;
	int booted, wf_pos;
	float pwr_scale, pwr_offset, histo_scale, histo_offset;
};

extern "C" {

void ref_set_binding(int portable) { g_binding = portable ? 1 : 0; }

void ref_fft(int n, const float *in, float *out, const float *win, int n_spectra)
{
	launch_fft(n, in, out, win, n_spectra);
}

/* one display launch on caller-owned buffers (N=1024 geometry of cl.c) */
void ref_display(const float *fft, int batch, float *wf, int wf_rows, int wf_offset,
                 float *hist, float t0r, float t0d, float hs, float ho,
                 float *spectrum, float alpha)
{
	Img iwf = { wf, 1024, wf_rows }, ih = { hist, 1024, 128 };
	k_disp.fft = (const float2 *)fft; k_disp.log2len = 10; k_disp.batch = (unsigned)batch;
	k_disp.wf = &iwf; k_disp.wf_offset = (unsigned)wf_offset; k_disp.hist = &ih;
	k_disp.t0r = t0r; k_disp.t0d = t0d; k_disp.hs = hs; k_disp.ho = ho;
	k_disp.spec = (float2 *)spectrum; k_disp.alpha = alpha;
	launch_display(1024);
}

ref_state *ref_new(void)
{
	ref_state *st = new ref_state();
	st->wf.assign(1024 * 1024, 0.0f);
	st->hist.assign(1024 * 128, 0.0f);
	st->spectrum.assign(4096, 0.0f);
	st->booted = 0; st->wf_pos = 0;
	/* fosphor.c:108-121 default window */
	for (int i = 0; i < 1024; i++) {
		float ft = 1024.0f, fp = (float)i;
		st->win[i] = (0.54f - 0.46f * cosf((2.0f * 3.141592f * fp) / ft)) * 1.855f;
	}
	/* fosphor.c:131-152 with (0, 10) */
	{
		int db0 = 0 - 10 * 10, db1 = 0;
		float k = log10f(1024.0f);
		st->pwr_offset = -(k + ((float)db0 / 20.0f));
		st->pwr_scale = 20.0f / (float)(db1 - db0);
		st->histo_scale = st->pwr_scale * 128.0f;	/* cl.c:1087 */
		st->histo_offset = st->pwr_offset;
	}
	return st;
}
void ref_free(ref_state *st) { delete st; }

void ref_set_window(ref_state *st, const float *win) { memcpy(st->win, win, sizeof(st->win)); }

void ref_set_power_range(ref_state *st, int db_ref, int db_per_div)
{
	int db0 = db_ref - 10 * db_per_div, db1 = db_ref;
	float k = log10f(1024.0f);
	st->pwr_offset = -(k + ((float)db0 / 20.0f));
	st->pwr_scale = 20.0f / (float)(db1 - db0);
	st->histo_scale = st->pwr_scale * 128.0f;
	st->histo_offset = st->pwr_offset;
}

/* cl.c:870-968.  strict: enforce the host-side batch cap (cl.c:885). */
int ref_process(ref_state *st, const float *samples, int len, int strict)
{
	if (len <= 0 || (len & (16 * 1024 - 1))) return -EINVAL;
	if (strict && len > 1024 * 1024) return -EINVAL;
	int batch = len / 1024;
	st->fft_out.resize((size_t)len * 2);
	launch_fft(1024, samples, st->fft_out.data(), st->win, batch);
	if (!st->booted) {
		float nf = -st->pwr_offset;
		std::fill(st->spectrum.begin(), st->spectrum.end(), nf);
		std::fill(st->wf.begin(), st->wf.end(), nf);
		std::fill(st->hist.begin(), st->hist.end(), 0.0f);
		st->booted = 1;
	}
	ref_display(st->fft_out.data(), batch, st->wf.data(), 1024, st->wf_pos, st->hist.data(),
	            16.0f, 1024.0f, st->histo_scale, st->histo_offset, st->spectrum.data(), 0.002f);
	st->wf_pos = (st->wf_pos + batch) & 1023;
	return 0;
}

float *ref_waterfall(ref_state *st) { return st->wf.data(); }
float *ref_histogram(ref_state *st) { return st->hist.data(); }
float *ref_spectrum(ref_state *st)  { return st->spectrum.data(); }
float *ref_fft_out(ref_state *st)   { return st->fft_out.data(); }
int    ref_waterfall_pos(ref_state *st) { return st->wf_pos; }
float  ref_histo_scale(ref_state *st)  { return st->histo_scale; }
float  ref_histo_offset(ref_state *st) { return st->histo_offset; }

} // extern "C"
