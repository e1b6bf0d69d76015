/*
 * fosphor_api.cpp -- C ABI of libfosphor_amd.so (include/fosphor.h, include/fosphor_amd.h)
 *
 * Host runtime that replaces lib/fosphor/cl.c + the compute half of
 * lib/fosphor/fosphor.c: device buffers, lazy window upload (cl.c:889-900),
 * first-run fills (cl.c:406-465), per-call launch sequence (cl.c:903-954),
 * waterfall ring position (cl.c:954,1073-1079), histogram range (cl.c:1081-1089),
 * tri-state BOOTING/PENDING/READY (cl.c:92-96), errno-style returns.
 *
 * There is no CPU fallback: without a HIP device fosphor_init() fails loudly.
 */
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "fosphor_internal.h"
#include "../../include/fosphor_amd.h"
#include "../../include/fosphor_amd_sink.h"
#include "../../include/fosphor_portable_math.h"

using namespace fosphor_amd;

#define FOSPHOR_AMD_VERSION "fosphor_amd 0.1 gfx950"

#define HIP_TRY(expr, what) do { \
		hipError_t _e = (expr); \
		if (_e != hipSuccess) { \
			fprintf(stderr, "[!] fosphor_amd: %s: %s\n", what, hipGetErrorString(_e)); \
			goto error; \
		} \
	} while (0)

enum { ST_BOOTING = 0, ST_PENDING = 1, ST_READY = 2 };	/* cl.c:92-96 */

static const int kMaxN = 65536;
static const int kSets = 5;		/* intermediate (bin index / partial) sets in rotation */
static const int kMaxK1Streams = 4;	/* `stream` + up to three more FFT streams */
/* N = 8192, space sharing: the FFT kernel (one work-group per CU, every register of it) takes 224 CUs when a launch's tiles divide
 * evenly among them, and the count / merge kernels of the launch before run on the 32 it leaves free -- the dispatcher fills CUs it
 * finds empty, no CU mask involved (hardware masks that remove CUs unevenly from the shader engines unbalance a grid of CU-sized
 * work-groups: tools/ubench/cu_mask_big.hip).  Measured at BASELINE C3: 333 -> 367-371 GSamples/s with 28 batches per call, 358
 * with 14, 345 with 7; 232 / 240 CUs leave the tail too little (it becomes the longer side), 216: 356 (DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8). */
#ifndef K1W_SHARE_CUS
#define K1W_SHARE_CUS 224		/* (A/B builds: 208 / 240 with 26 / 30 batches per call, profiles/r06_c3.md) */
#endif
static const int kK1wShareCus = K1W_SHARE_CUS;
static const int kSubSamplesLog2 = 26;	/* default sub-launch: 64 Mi samples (64 reference batches of 1024 x 1024) */

static const int kRiseMax = 8192;	/* largest batch served by the rise/decay table */

struct fosphor
{
	/* geometry / constants */
	int log2n, n, n_bins, wf_rows;
	int bins16;				/* bin indices are 16-bit (2 spectra per dword) */
	int tw_len, tw_off[8];
	float t0r, t0d, alpha;
	int max_spectra, max_batches;
	int iq_half;				/* device IQ is fp16 pairs (N = 65536 only) */
	size_t stage_samples;			/* capacity of one host staging slot, samples */

	/* reference-visible settings (private.h:44-54) */
	float fft_win[kMaxN];
	int   win_dirty;
	struct { int db_ref, db_per_div; float scale, offset; } power;
	struct { double center, span; } frequency;
	float histo_scale, histo_offset;	/* cl.c:1087-1088 */
	int   thr_dirty;

	/* device */
	int device;
	hipStream_t stream;
	int own_stream;
	float    *d_win;
	float2   *d_tw;
	double   *d_thr;
	float    *d_wf_pp[2], *d_hist;		/* two waterfall rings: a call that rewrites every row targets the other one,
						 * so its K1s need not wait for the previous call's (see run()) */
	int       wf_cur;			/* ring the results are in */
	hipEvent_t ev_wf[2];			/* last K1 that stored rows into ring b */
	int       wf_used[2];
	hipStream_t wf_stream[2];
	float2   *d_spectrum;
	uint32_t *d_bins_pp[kSets];		/* rotating sets: K1 of launch i+1 overlaps K2/K3 of launch i */
	float2   *d_partial_pp[kSets];
	uint32_t *d_bins;			/* current set */
	float2   *d_partial;
	int       pp;
	hipStream_t k1_streams[kMaxK1Streams];	/* [0] = `stream`; the K1s of consecutive sub-launches of a device-resident call rotate over
						 * n_k1_streams of them: the next K1s are already queued when work-groups of the current one exit */
	int       n_sets;			/* intermediate sets in use (3, <= kSets) */
	int       n_k1_streams;			/* FOSPHOR_AMD_K1_STREAMS (default 2) */
	hipEvent_t ev_k1s_done[kMaxK1Streams];
	int       alt;				/* FOSPHOR_AMD_ALT=0 keeps every K1 on `stream` */
	int       k1_seq;
	int       relaxed;			/* fosphor_amd_set_input_ordering(self, 0) */
	hipEvent_t ev_in;
	long long sub_samples;			/* samples per sub-launch of a device-resident call */
	hipStream_t stream2;			/* K2 (and K3 unless pipe3) of the multi-batch path */
	hipStream_t stream3;			/* K3 of the multi-batch path: K3 of launch i beside K2 of launch i+1 */
	int       pipe3;			/* FOSPHOR_AMD_PIPE3=0 keeps K3 on stream2 */
	int       hset;				/* hit-count / live-sum set of the next launch (two, like the bin sets) */
	int       hset_used[2];
	hipEvent_t ev_k2_done[2];		/* K2 wrote hit-count set h */
	hipEvent_t ev_h_free[2];		/* K3 finished reading hit-count set h */
	hipEvent_t ev_k3_done;			/* orders K3s that are issued on different streams */
	hipEvent_t ev_tail;			/* N = 8192: behind the merge kernel of the last launch (is there a tail to share the chip with?) */
	int       tail_set;
	hipStream_t last_k3_stream;
	hipStream_t k2_stream_last;		/* stream of the most recent count kernel */
	hipEvent_t ev_k1_done[kSets];		/* K1 wrote set pp */
	hipEvent_t ev_set_free[kSets];		/* K2 finished reading set pp */
	int       set_used[kSets];
	int       overlap;			/* 1: two-stream pipeline for process paths */
	int       k1_variant;			/* FOSPHOR_AMD_K1: 1 (default) = wave per spectrum, 2 = two waves per spectrum (what odd hops use) */
	/* tuning / test knobs, read from the environment ONCE, at init (nothing on the submit path calls getenv) */
	int       kn_tile;			/* FOSPHOR_AMD_TILE: spectra per tile, 0 = pick_tile's choice */
	int       kn_rowmask_off;		/* FOSPHOR_AMD_ROWMASK=0: dense count hand-off at N = 65536 */
	int       kn_no_sum16;			/* FOSPHOR_AMD_NO_SUM16 */
	int       kn_frame_group;		/* FOSPHOR_AMD_FRAME_GROUP: chunks per count work-group of a sharded frame (default 4) */
	int       kn_k1w_share_off;		/* FOSPHOR_AMD_K1W_SHARE=0: no space sharing at N = 8192 */
	int       n_cus;			/* hipDeviceProp_t::multiProcessorCount */
	int       share_cus;			/* N = 8192: work-groups of an FFT launch that leaves CUs to the count / merge kernels (0: never) */
	long long k1w_shared, k1w_full;		/* N = 8192: FFT launches made in the shared / the full-chip form (fosphor_amd_share_stats) */
	uint32_t *d_hc;
	uint32_t *d_hc_export;			/* [n_bins][N] last batch, written by K3 on the 16-bit path */
	const uint16_t *export_src;		/* ... made from the last batch's slabs when fosphor_amd_get_buffers asks */
	const uint32_t *export_mask;
	hipStream_t export_stream;		/* the stream the K2 that wrote export_src ran on */
	uint32_t *d_rowmask;			/* K2 -> K3: one bit per (batch, slab, bin row) "this row has counts and is stored"; 2 sets */
	int       mask_words;			/* ceil(n_bins / 32) */
	uint8_t  *d_hot;			/* [N/64][n_bins]: some cell of the row is above the fast-exit level (K3 maintains it) */
	uint32_t *d_rowlist;			/* [2 + rows]: the live rows of a merge (sparse form) between two counts used alternately */
	int       rowlist_flip;
	int       hot_valid;			/* 0 after anything but the 16-bit K3 wrote the histogram */
	uint16_t *d_slab16;			/* per-chunk packed 16-bit count slabs of batches longer than 1024 spectra / of a shard */
	int       slab_chunks;			/* capacity of d_slab16 in 1024-spectrum chunks */
	int       last_hc16;
	float    *d_live_sum, *d_vmax;
	float    *d_chunk_sum, *d_chunk_max;	/* [max_spectra/16][N] */
	long long *d_dbg;			/* K1_TIMING builds only (FOSPHOR_AMD_K1_TIMING=1) */
	uint32_t *d_palette;			/* colour-map scratch (fosphor_cmap.hip), allocated on first use */
	float2   *d_rise;			/* [kRiseMax+1] (d, e) per hit count */
	float2   *h_rise;			/* pinned */
	int       rise_batch;			/* batch the table was built for (0 = none) */
	float     rise_t0r, rise_t0d;
	int       slot;				/* partial-array slot used by accumulate/merge */
	float2   *d_scratch;			/* N = 65536: [max_spectra][N] spectrum between the two FFT stages */
	int       k1h_fused;			/* N = 65536: both levels of the plan in one kernel, the intermediate in the XCDs' L2 (always 1 since round 4) */
	uint32_t *d_k1h_sync;			/* its cluster counters */
	uint32_t *h_k1h_err;			/* ... and its error word (host memory the kernel writes: work-groups of a cluster on different XCDs) */

	/* host->device staging for fosphor_process (pinned ring of 2) */
	float2   *h_stage[2];
	float2   *d_stage[2];
	hipEvent_t stage_free[2];
	hipEvent_t upload_done;			/* H2D of the last fosphor_amd_process_pinned */
	hipStream_t copy_stream;		/* ... queued here, so that the upload of one batch runs beside the kernels of the batch before
						 * (created by the first fosphor_amd_process_pinned) */
	hipEvent_t upload_slot[2];		/* upload into d_stage[k] done: the instance's stream waits for it before the FFT kernel */
	int       stage_used[2];
	size_t    d_stage_cap[2];		/* samples d_stage[k] holds */
	int       pend_slot[2], pend_len[2], pend_head, pend_n;	/* uploads queued by fosphor_amd_upload_pinned, kernels not yet */
	int       stage_idx;
	double   *h_thr;			/* pinned, n_bins+1 */
	float    *h_win;			/* pinned, N */

	/* state */
	int state;
	int wf_pos;
	int last_batches;			/* batches in the most recent launch (hitcount view) */
	int last_slot0;

	/* the rise/decay table serves batches up to kRiseMax (K3's 16-bit path needs it) */
	bool rise_ok(int batch) const { return batch <= 8192; }

	/* profiling */
	int prof;				/* 0 off, 1 every kernel, 2 K1 only (fewer events beside K2/K3) */
	int prof_open;
	std::vector<hipEvent_t> ev_pool;
	std::vector<int> ev_kind;		/* kernel index per (start, stop) pair */
	size_t ev_used;
	std::vector<hipEvent_t> xev_pool;	/* (start, stop) pairs around the exchanges of the multi-GPU path, while profiling is on */
	size_t xev_used;
};

/* ------------------------------------------------------------------------ */
/* Host-side tables                                                         */
/* ------------------------------------------------------------------------ */

/* Twiddles.  N = 1024: exactly as the reference forms them (fft.cl:62-68,162-166,286-297), native_sin / native_cos pinned to
 * fosphor_portable_math.h; one block per radix-8 pass with p = 8, 64 ([k < p][n = 1..7]), then the final radix-2 pass
 * ([k < N/2]): the kTw2Off / kTw3Off / kTw4Off layout of fosphor_internal.h.
 *
 * N = 8192 and N = 65536 run this build's own plans (no reference behaviour exists at these lengths; oracle/fosphor_oracle.c,
 * o_pass_radix8_fma / o_pass_radix16_fma / o_pass_radix2_fma, restates them): the reference's Stockham passes with the twiddles
 * ON the radix-2 butterflies of a pass instead of on its inputs, so an item needs the twiddles of its stages -- every one of them
 * exp(-j pi q / den) for integers q, den, formed like the reference's (one float expression, pinned sin / cos): tw_long().
 *   both:       block  0    the constants W16, W8, W16^3 (first pass)
 *   N = 8192:   blocks 1-2  passes p = 16, 256: [k < p][8] = w^8, w^4, w^2, w^2 W8, w, w W16, w W8, w W16^3   (w = exp(-j pi k / (8 p)))
 *               block  3    the radix-2 pass: [k < 4096] = exp(-j pi k / 4096)
 *   N = 65536:  blocks 1-3  passes p = 16, 256, 4096: [k < p][8], as above */
static float2 tw_long(int q, int den)
{
	const float PI_F = 3.141592653589f;		/* fft.cl:26 */
	const float arg = -PI_F * (float)q / (float)den;
	return make_float2(fpm_cosf(arg), fpm_sinf(arg));
}

static int build_twiddles16(float2 *tw, int *offsets /* [8] or NULL */)
{
	int pos = 0, q = 0;
	if (offsets) offsets[q] = pos;
	q++;
	for (int m = 1; m <= 3; m++) {
		if (tw) tw[pos] = tw_long(m, 8);
		pos++;
	}
	for (int p = 16; p <= 4096; p *= 16, q++) {
		if (offsets) offsets[q] = pos;
		const int d = 8 * p;
		for (int k = 0; k < p; k++) {
			const int qs[8] = { 8 * k, 4 * k, 2 * k, 2 * k + 2 * p, k, k + p, k + 2 * p, k + 3 * p };
			for (int f = 0; f < 8; f++) {
				if (tw) tw[pos] = tw_long(qs[f], d);
				pos++;
			}
		}
	}
	return pos;
}

static int build_twiddles13(float2 *tw, int *offsets /* [8] or NULL */)
{
	const int n = 8192;
	int pos = 0, q = 0;
	if (offsets) offsets[q] = pos;
	q++;
	for (int m = 1; m <= 3; m++) {
		if (tw) tw[pos] = tw_long(m, 8);
		pos++;
	}
	for (int p = 16; p <= 256; p *= 16, q++) {
		if (offsets) offsets[q] = pos;
		const int d = 8 * p;
		for (int k = 0; k < p; k++) {
			const int qs[8] = { 8 * k, 4 * k, 2 * k, 2 * k + 2 * p, k, k + p, k + 2 * p, k + 3 * p };
			for (int f = 0; f < 8; f++) {
				if (tw) tw[pos] = tw_long(qs[f], d);
				pos++;
			}
		}
	}
	if (offsets) offsets[q] = pos;
	for (int k = 0; k < n / 2; k++) {
		if (tw) tw[pos] = tw_long(k, n / 2);
		pos++;
	}
	return pos;
}

static int build_twiddles(float2 *tw, int log2n, int *offsets /* [8] or NULL */)
{
	const float PI_F = 3.141592653589f;		/* fft.cl:26 */
	const int n = 1 << log2n;
	int pos = 0, q = 0;
	if (log2n == 16)
		return build_twiddles16(tw, offsets);
	if (log2n == 13)
		return build_twiddles13(tw, offsets);
	for (int p = 8; p < n / 2; p *= 8, q++) {
		if (offsets) offsets[q] = pos;
		for (int k = 0; k < p; k++) {
			const float alpha = -PI_F * (float)k / (float)(4 * p);
			for (int f = 1; f < 8; f++) {
				const float arg = (float)f * alpha;
				if (tw) tw[pos] = make_float2(fpm_cosf(arg), fpm_sinf(arg));
				pos++;
			}
		}
	}
	if (offsets) offsets[q] = pos;
	for (int k = 0; k < n / 2; k++) {
		const float alpha = -PI_F * (float)k / (float)(n / 2);
		const float arg = (float)1 * alpha;
		if (tw) tw[pos] = make_float2(fpm_cosf(arg), fpm_sinf(arg));
		pos++;
	}
	return pos;
}

/* Exact bin thresholds on the double squared magnitude: thr[b] = smallest s >= 0 with
 * oracle bin(s) >= b.  bin(s) is monotone while hypot(s) is finite (checked exhaustively by
 * oracle/tools/pm_check.c); thr[nb] = smallest s whose hypot overflows float. */
static void build_thresholds(double *thr, int nb, float hs, float ho)
{
	/* largest s with finite (float)sqrt(s) */
	uint64_t lo = 0, hi = fpm_d2u(1.0e80);
	while (hi - lo > 1) {
		uint64_t mid = lo + (hi - lo) / 2;
		if (isinf(fpm_hypot_from_sqmag(fpm_u2d(mid)))) hi = mid; else lo = mid;
	}
	const uint64_t s_inf = hi;
	thr[0] = -1.0;
	thr[nb] = fpm_u2d(s_inf);
	for (int b = 1; b < nb; b++) {
		/* invariant: bin(lo) < b <= bin(hi) on [0, s_inf) */
		if (fpm_bin_from_sqmag(fpm_u2d(s_inf - 1), hs, ho, nb) < b) {
			thr[b] = fpm_u2d(s_inf);	/* unreachable bin */
			continue;
		}
		lo = 0; hi = s_inf - 1;
		while (hi - lo > 1) {
			uint64_t mid = lo + (hi - lo) / 2;
			if (fpm_bin_from_sqmag(fpm_u2d(mid), hs, ho, nb) >= b) hi = mid; else lo = mid;
		}
		thr[b] = fpm_u2d(hi);
	}
}

/* ------------------------------------------------------------------------ */
/* Init / release                                                           */
/* ------------------------------------------------------------------------ */

/* Events that order kernels of this device among the instance's own streams (and the timing
 * events) need no system-scope fence: a default event makes the queue write back and invalidate
 * the L2s when it is recorded, which costs microseconds between dependent kernels and sends the
 * consumer's reads to HBM.  Kernel boundaries keep their device-scope release/acquire.
 * (Events the HOST waits on for host-visible data -- staging, upload -- keep the default.) */
/* Output streams of the 65536-point kernel (waterfall rows, bin indices) live in UNCACHED device memory: its clusters keep their
 * intermediate in the XCD's L2 and every byte streamed through that L2 pushes intermediate lines out, to be written back to HBM although
 * they are dead.  Uncached stores go past the L2: WRITE_SIZE per 1024-spectrum frame 649 -> 541 MiB at the same kernel time (DESIGN.md
 * section 8; A/B'd in round 4 through an environment switch that has since gone).  The other configurations have no such hot set and keep plain memory (the 8192-point
 * kernel's index stores run 6 % slower uncached). */
static hipError_t alloc_output(void **p, size_t bytes, bool uncached)
{
	if (uncached && hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached) == hipSuccess)
		return hipSuccess;
	return hipMalloc(p, bytes);
}

static unsigned dep_event_flags(void)
{
	/* (no system fence: the events order kernels of this device among its own streams, nothing the host or another device reads) */
	return hipEventDisableTiming | hipEventDisableSystemFence;
}

extern "C" const char *fosphor_amd_version(void) { return FOSPHOR_AMD_VERSION; }

extern "C" void fosphor_release(struct fosphor *self)
{
	if (!self)
		return;
	if (self->stream)
		(void)hipStreamSynchronize(self->stream);
	(void)hipFree(self->d_win); (void)hipFree(self->d_tw); (void)hipFree(self->d_thr);
	(void)hipFree(self->d_wf_pp[0]); (void)hipFree(self->d_wf_pp[1]);
	(void)hipFree(self->d_hist); (void)hipFree(self->d_spectrum);
	for (int i = 0; i < 2; i++)
		if (self->ev_wf[i]) (void)hipEventDestroy(self->ev_wf[i]);
	if (self->ev_in) (void)hipEventDestroy(self->ev_in);
	for (int i = 1; i < kMaxK1Streams; i++) {
		if (self->k1_streams[i]) { (void)hipStreamSynchronize(self->k1_streams[i]); (void)hipStreamDestroy(self->k1_streams[i]); }
		if (self->ev_k1s_done[i]) (void)hipEventDestroy(self->ev_k1s_done[i]);
	}
	for (int i = 0; i < kSets; i++) {
		(void)hipFree(self->d_bins_pp[i]); (void)hipFree(self->d_partial_pp[i]);
		if (self->ev_k1_done[i]) (void)hipEventDestroy(self->ev_k1_done[i]);
		if (self->ev_set_free[i]) (void)hipEventDestroy(self->ev_set_free[i]);
	}
	if (self->copy_stream) { (void)hipStreamSynchronize(self->copy_stream); (void)hipStreamDestroy(self->copy_stream); }
	if (self->stream2) { (void)hipStreamSynchronize(self->stream2); (void)hipStreamDestroy(self->stream2); }
	if (self->stream3) { (void)hipStreamSynchronize(self->stream3); (void)hipStreamDestroy(self->stream3); }
	for (int i = 0; i < 2; i++) {
		if (self->ev_k2_done[i]) (void)hipEventDestroy(self->ev_k2_done[i]);
		if (self->ev_h_free[i]) (void)hipEventDestroy(self->ev_h_free[i]);
	}
	if (self->ev_k3_done) (void)hipEventDestroy(self->ev_k3_done);
	if (self->ev_tail) (void)hipEventDestroy(self->ev_tail);
	(void)hipFree(self->d_hc); (void)hipFree(self->d_hc_export); (void)hipFree(self->d_slab16);
	(void)hipFree(self->d_rowmask); (void)hipFree(self->d_hot); (void)hipFree(self->d_rowlist);
	(void)hipFree(self->d_live_sum); (void)hipFree(self->d_vmax);
	(void)hipFree(self->d_chunk_sum); (void)hipFree(self->d_chunk_max);
	(void)hipFree(self->d_rise);
	(void)hipFree(self->d_palette);
	(void)hipFree(self->d_scratch);
	(void)hipFree(self->d_k1h_sync);
	if (self->h_k1h_err) (void)hipHostFree(self->h_k1h_err);
	(void)hipFree(self->d_dbg);
	if (self->h_rise) (void)hipHostFree(self->h_rise);
	for (int i = 0; i < 2; i++) {
		if (self->h_stage[i]) (void)hipHostFree(self->h_stage[i]);
		if (self->d_stage[i]) (void)hipFree(self->d_stage[i]);
		if (self->stage_free[i]) (void)hipEventDestroy(self->stage_free[i]);
	}
	if (self->upload_done) (void)hipEventDestroy(self->upload_done);
	for (int i = 0; i < 2; i++)
		if (self->upload_slot[i]) (void)hipEventDestroy(self->upload_slot[i]);
	if (self->h_thr) (void)hipHostFree(self->h_thr);
	if (self->h_win) (void)hipHostFree(self->h_win);
	for (hipEvent_t e : self->ev_pool) (void)hipEventDestroy(e);
	for (hipEvent_t e : self->xev_pool) (void)hipEventDestroy(e);
	if (self->own_stream && self->stream)
		(void)hipStreamDestroy(self->stream);
	delete self;
}

extern "C" struct fosphor *fosphor_amd_init(const struct fosphor_amd_config *cfg)
{
	struct fosphor *self = new (std::nothrow) fosphor();
	int ndev = 0;
	size_t tiles_max;
	std::vector<float2> tw;
	int cu_reserved = 0;
	uint32_t cu_mask_fft[8], cu_mask_cnt[8];

	if (!self)
		return NULL;

	self->log2n   = (cfg && cfg->fft_len_log > 0) ? cfg->fft_len_log : kLog2N;
	self->n       = 1 << self->log2n;
	self->n_bins  = (cfg && cfg->n_bins > 0) ? cfg->n_bins : 128;
	self->wf_rows = (cfg && cfg->wf_rows > 0) ? cfg->wf_rows : 1024;
	self->t0r     = (cfg && cfg->t0r > 0.0f) ? cfg->t0r : 16.0f;		/* cl.c:714 */
	self->t0d     = (cfg && cfg->t0d > 0.0f) ? cfg->t0d : 1024.0f;		/* cl.c:715 */
	self->alpha   = (cfg && cfg->alpha > 0.0f) ? cfg->alpha : 0.002f;	/* cl.c:716 */
	self->max_spectra = (cfg && cfg->max_spectra > 0) ? cfg->max_spectra : 1024;
	self->max_batches = (cfg && cfg->max_batches > 0) ? cfg->max_batches
	                    : (self->max_spectra / 1024 > 8 ? self->max_spectra / 1024 : 8);

	if (self->log2n != 10 && self->log2n != 13 && self->log2n != 16) {
		fprintf(stderr, "[!] fosphor_amd: fft_len_log=%d not supported (10, 13 or 16)\n", self->log2n);
		goto error;
	}
	self->iq_half = (cfg && cfg->iq_format == FOSPHOR_AMD_IQ_FP16);
	if (cfg && cfg->iq_format != FOSPHOR_AMD_IQ_FP32 && cfg->iq_format != FOSPHOR_AMD_IQ_FP16) {
		fprintf(stderr, "[!] fosphor_amd: iq_format=%d unknown\n", cfg->iq_format);
		goto error;
	}
	if (self->iq_half && self->log2n != 16) {
		fprintf(stderr, "[!] fosphor_amd: fp16 IQ is only implemented for fft_len_log=16\n");
		goto error;
	}
	if (self->n_bins < 16 || self->n_bins > 512 || (self->n_bins & 15)) {
		fprintf(stderr, "[!] fosphor_amd: n_bins=%d not supported (16..512, multiple of 16)\n", self->n_bins);
		goto error;
	}
	/* 8-bit bin indices (4 spectra per dword) need n_bins <= 256 and the 1024-point kernels */
	self->bins16 = (self->n_bins > 256) || (self->log2n != 10);
	if (self->wf_rows & (self->wf_rows - 1)) {
		fprintf(stderr, "[!] fosphor_amd: wf_rows=%d is not a power of two\n", self->wf_rows);
		goto error;
	}
	if (self->max_spectra & 15) {
		fprintf(stderr, "[!] fosphor_amd: max_spectra=%d is not a multiple of 16\n", self->max_spectra);
		goto error;
	}

	{
		hipError_t e = hipGetDeviceCount(&ndev);
		if (e != hipSuccess || ndev < 1) {
			fprintf(stderr, "[!] fosphor_amd: no HIP device available (%s, %d devices); this library has no CPU path\n",
			        hipGetErrorString(e), ndev);
			goto error;
		}
	}
	if (cfg && cfg->device >= 0) {
		HIP_TRY(hipSetDevice(cfg->device), "hipSetDevice");
		self->device = cfg->device;
	} else {
		HIP_TRY(hipGetDevice(&self->device), "hipGetDevice");
	}
	{
		hipDeviceProp_t prop;
		HIP_TRY(hipGetDeviceProperties(&prop, self->device), "hipGetDeviceProperties");
		self->n_cus = prop.multiProcessorCount;
		/* space sharing is laid out for the whole MI355X (256 CUs: 224 + 32); any other CU count -- a partitioned part -- runs
		 * every launch in the full-chip form */
		self->share_cus = (self->n_cus == 256) ? kK1wShareCus : 0;
	}
	{
		const char *e;
		e = getenv("FOSPHOR_AMD_TILE");        self->kn_tile = e ? atoi(e) : 0;
		e = getenv("FOSPHOR_AMD_ROWMASK");     self->kn_rowmask_off = (e && *e == '0');
		self->kn_no_sum16 = getenv("FOSPHOR_AMD_NO_SUM16") != NULL;
		e = getenv("FOSPHOR_AMD_FRAME_GROUP"); self->kn_frame_group = e ? atoi(e) : 4;
		e = getenv("FOSPHOR_AMD_K1W_SHARE");   self->kn_k1w_share_off = (e && *e == '0');
	}

	/* Measurement only, probe builds (-DFOSPHOR_AMD_PROBES; profiles/r04_ceiling.md): FOSPHOR_AMD_DBG_CUMASK=k reserves k CUs (k / 8 per XCD; mask bit i is CU i / 8 of
	 * XCD i % 8, tools/ubench/cu_mask_probe.hip) for the count / merge streams and confines every FFT stream -- including the
	 * instance's main stream, which is then the library's own and not the caller's -- to the rest.  Space-sharing by mask
	 * measured far worse than the hardware's own interleaving (DESIGN_HISTORY.md 4a); nothing in the product path sets it. */
#ifdef FOSPHOR_AMD_PROBES
	{
		const char *e = getenv("FOSPHOR_AMD_DBG_CUMASK");
		cu_reserved = e ? atoi(e) : 0;
		if (cu_reserved < 8 || cu_reserved > 128 || (cu_reserved & 7))
			cu_reserved = 0;
	}
#endif
	for (int w = 0; w < 8; w++) {
		const int lo = 32 * w, split = 256 - cu_reserved;
		cu_mask_fft[w] = split >= lo + 32 ? ~0u : (split <= lo ? 0u : ((1u << (split - lo)) - 1u));
		cu_mask_cnt[w] = ~cu_mask_fft[w];
	}
	if (cfg && cfg->stream && !cu_reserved) {
		self->stream = (hipStream_t)cfg->stream;
	} else {
		if (cu_reserved)
			HIP_TRY(hipExtStreamCreateWithCUMask(&self->stream, 8, cu_mask_fft), "hipExtStreamCreateWithCUMask");
		else
			HIP_TRY(hipStreamCreateWithFlags(&self->stream, hipStreamNonBlocking), "hipStreamCreate");
		self->own_stream = 1;
	}

	tiles_max = (size_t)self->max_spectra / 4;
	HIP_TRY(hipMalloc((void **)&self->d_win, sizeof(float) * self->n), "alloc window");
	self->tw_len = build_twiddles(NULL, self->log2n, NULL);
	HIP_TRY(hipMalloc((void **)&self->d_tw, sizeof(float2) * self->tw_len), "alloc twiddles");
	HIP_TRY(hipMalloc((void **)&self->d_thr, sizeof(double) * (self->n_bins + 1)), "alloc thresholds");
	for (int i = 0; i < 2; i++) {
		HIP_TRY(alloc_output((void **)&self->d_wf_pp[i], sizeof(float) * (size_t)self->wf_rows * self->n, self->log2n == 16), "alloc waterfall");
		HIP_TRY(hipEventCreateWithFlags(&self->ev_wf[i], dep_event_flags()), "create event");
	}
	HIP_TRY(hipEventCreateWithFlags(&self->ev_in, dep_event_flags()), "create event");
	/* Only the streams in use are created: the runtime spreads streams over a few hardware queues (4 unless
	 * GPU_MAX_HW_QUEUES says otherwise) and two streams that share one do not overlap. */
	{
		const char *e = getenv("FOSPHOR_AMD_K1_STREAMS");
		self->n_k1_streams = (e && atoi(e) >= 1 && atoi(e) <= kMaxK1Streams) ? atoi(e) : 2;
	}
	self->k1_streams[0] = self->stream;
	for (int i = 1; i < self->n_k1_streams; i++) {
		if (cu_reserved)
			HIP_TRY(hipExtStreamCreateWithCUMask(&self->k1_streams[i], 8, cu_mask_fft), "hipExtStreamCreateWithCUMask (FFT stream)");
		else
			HIP_TRY(hipStreamCreateWithFlags(&self->k1_streams[i], hipStreamNonBlocking), "hipStreamCreate (FFT stream)");
		HIP_TRY(hipEventCreateWithFlags(&self->ev_k1s_done[i], dep_event_flags()), "create event");
	}
	HIP_TRY(hipMalloc((void **)&self->d_hist, sizeof(float) * (size_t)self->n_bins * self->n), "alloc histogram");
	HIP_TRY(hipMalloc((void **)&self->d_spectrum, sizeof(float2) * 2 * self->n), "alloc spectrum");
	for (int i = 0; i < kSets; i++) {
		HIP_TRY(alloc_output((void **)&self->d_bins_pp[i], (size_t)self->max_spectra * self->n * (self->bins16 ? 2 : 1), self->log2n == 16), "alloc bin indices");
		HIP_TRY(hipMalloc((void **)&self->d_partial_pp[i], sizeof(float2) * tiles_max * self->n), "alloc partials");
		HIP_TRY(hipEventCreateWithFlags(&self->ev_k1_done[i], dep_event_flags()), "create event");
		HIP_TRY(hipEventCreateWithFlags(&self->ev_set_free[i], dep_event_flags()), "create event");
	}
	self->d_bins = self->d_bins_pp[0];
	self->d_partial = self->d_partial_pp[0];
	if (self->log2n == 16) {
		/* One FFT kernel takes a spectrum through both 256-point levels of the radix-16 plan, the intermediate staying in the
		 * XCD's L2 (DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8; the two-kernel form of rounds 1-3 went with the radix-8 plan in round 4). */
		self->k1h_fused = 1;
		/* 512 KiB of intermediate per cluster, at most 8 clusters on each of the 8 XCDs */
		HIP_TRY(hipMalloc((void **)&self->d_scratch, sizeof(float2) * (size_t)64 * self->n), "alloc cluster intermediates");
		HIP_TRY(hipMalloc((void **)&self->d_k1h_sync, sizeof(uint32_t) * 64 * 64), "alloc cluster counters");
		HIP_TRY(hipMemset(self->d_k1h_sync, 0, sizeof(uint32_t) * 64 * 64), "clear cluster counters");
		HIP_TRY(hipHostMalloc((void **)&self->h_k1h_err, 64, hipHostMallocMapped), "alloc error word");
		self->h_k1h_err[0] = 0;
	}
	/* host staging slot: the reference's cap of 1024 spectra per call (cl.c:885), or this instance's */
	self->stage_samples = (size_t)self->n * (self->max_spectra < 1024 ? self->max_spectra : 1024);
	if (cu_reserved) {
		HIP_TRY(hipExtStreamCreateWithCUMask(&self->stream2, 8, cu_mask_cnt), "hipExtStreamCreateWithCUMask (count stream)");
		HIP_TRY(hipExtStreamCreateWithCUMask(&self->stream3, 8, cu_mask_cnt), "hipExtStreamCreateWithCUMask (merge stream)");
	} else {
		HIP_TRY(hipStreamCreateWithFlags(&self->stream2, hipStreamNonBlocking), "hipStreamCreate (count stream)");
		HIP_TRY(hipStreamCreateWithFlags(&self->stream3, hipStreamNonBlocking), "hipStreamCreate (merge stream)");
	}
	for (int i = 0; i < 2; i++) {
		HIP_TRY(hipEventCreateWithFlags(&self->ev_k2_done[i], dep_event_flags()), "create event");
		HIP_TRY(hipEventCreateWithFlags(&self->ev_h_free[i], dep_event_flags()), "create event");
	}
	HIP_TRY(hipEventCreateWithFlags(&self->ev_k3_done, dep_event_flags()), "create event");
	HIP_TRY(hipEventCreateWithFlags(&self->ev_tail, dep_event_flags()), "create event");
	if (getenv("FOSPHOR_AMD_K1_TIMING")) {
		HIP_TRY(hipMalloc((void **)&self->d_dbg, sizeof(long long) * 16 * 4 * kK1MaxBlocks), "alloc timing buffer");
		HIP_TRY(hipMemset(self->d_dbg, 0, sizeof(long long) * 16 * 4 * kK1MaxBlocks), "clear timing buffer");
	}
	{
		const char *e = getenv("FOSPHOR_AMD_OVERLAP");
		self->overlap = !(e && *e == '0');
		/* The fused 65536-point kernel fills every CU.  Rounds 2-3 therefore ran FFT -> count -> scan -> merge one after the other on
		 * ONE stream (then 134-136 -> 142-144 GSamples/s: the radix-8 kernel of 16 waves and 134 KiB of LDS crawled when count / merge
		 * work-groups reached the CUs first).  The radix-16 kernel of round 4 (8 waves) does not: with the streams, count on the
		 * second and merge on the third, the tail of frame f runs in the gaps of frame f + 1's FFT kernel -- its ramp-up, and the
		 * CUs its clusters leave as the tiles run out: 215.3 -> 229.5 GSamples/s (four interleaved runs each; two streams 224.5;
		 * with the FFT kernel made to wait for the previous merge 199.7).  FOSPHOR_AMD_OVERLAP=0: one stream. */
		/* N = 8192: the FFT kernel owns every CU's LDS and registers, so count and merge cannot run beside it either way; on one
		 * stream the kernel boundaries are cheaper than cross-stream events (measured 286.6 / 287.8 / 287.1 against 286.1 / 284.7 /
		 * 283.5 GSamples/s), and K1's busy time is no longer stretched by launches waiting for each other (0.363 vs 0.29-0.345). */
		/* (round 4: the streams are back for N = 8192 -- with space sharing, kK1wShareCus, count and merge DO run beside the next FFT
		 * launch; for launch shapes that cannot share, the streams cost 0.3 %: 326 against 327 GSamples/s.  FOSPHOR_AMD_OVERLAP=0:
		 * one stream.) */
		e = getenv("FOSPHOR_AMD_K1");
		self->k1_variant = (e && *e == '2') ? 2 : 1;
		e = getenv("FOSPHOR_AMD_PIPE3");
		self->pipe3 = e ? (*e == '1') : (self->log2n == 16);	/* (N = 65536: merge on its own stream, above) */
		e = getenv("FOSPHOR_AMD_ALT");
		self->alt = !(e && *e == '0');
		self->n_sets = 3;		/* (of kSets allocated: more in rotation measured nothing) */
		if (self->log2n == 13 && !getenv("FOSPHOR_AMD_K1_STREAMS"))
			self->n_k1_streams = 1;		/* a second FFT launch in flight would take the CUs left to count / merge */
		if (self->n_k1_streams == 1)
			self->alt = 0;
		e = getenv("FOSPHOR_AMD_SUB_LOG2");		/* tuning: log2 of the samples per sub-launch */
		/* (N = 8192: the one-work-group-per-CU FFT kernel owns the whole LDS, so K2 cannot run beside it and a
		 * smaller piece only adds serialised kernel boundaries: twice the default) */
		/* (round 4, N = 8192: 1 Gi samples -- up to 32 batches of 4096 spectra in one FFT launch: with space sharing a launch boundary
		 * costs ~40 us, during which the count kernels of the finished launch and the work-groups of the next FFT launch compete for
		 * the CUs) */
		self->sub_samples = 1LL << ((e && atoi(e) >= 14 && atoi(e) <= 34) ? atoi(e) : (self->log2n == 13 ? 30 : kSubSamplesLog2));
	}
	HIP_TRY(hipMalloc((void **)&self->d_hc, sizeof(uint32_t) * (size_t)self->max_batches * self->n_bins * self->n), "alloc hit counts");
	HIP_TRY(hipMalloc((void **)&self->d_hc_export, sizeof(uint32_t) * (size_t)self->n_bins * self->n), "alloc hit count view");
	self->mask_words = (self->n_bins + 31) / 32;
	HIP_TRY(hipMalloc((void **)&self->d_rowmask, sizeof(uint32_t) * 2 * (size_t)self->max_batches * (self->n / 64) * self->mask_words), "alloc row masks");
	HIP_TRY(hipMalloc((void **)&self->d_hot, (size_t)(self->n / 64) * self->n_bins), "alloc row flags");
	HIP_TRY(hipMemset(self->d_hot, 1, (size_t)(self->n / 64) * self->n_bins), "set row flags");
	HIP_TRY(hipMalloc((void **)&self->d_rowlist, sizeof(uint32_t) * (2 + (size_t)(self->n / 64) * self->n_bins)), "alloc row list");
	HIP_TRY(hipMemset(self->d_rowlist, 0, sizeof(uint32_t) * (2 + (size_t)(self->n / 64) * self->n_bins)), "clear row list");
	self->rowlist_flip = 0;
	self->hot_valid = 0;
	if (self->max_spectra > 1024) {
		/* one slab per 1024-spectrum chunk of the largest launch: a whole shard (accumulate) or a sub-launch */
		self->slab_chunks = self->max_spectra / 1024;
		HIP_TRY(hipMalloc((void **)&self->d_slab16, sizeof(uint16_t) * (size_t)self->slab_chunks * self->n_bins * self->n), "alloc count slabs");
	}
	/* two sets of max_batches slots (the 16-bit hit counts of the second set use the upper half of d_hc) */
	HIP_TRY(hipMalloc((void **)&self->d_live_sum, sizeof(float) * 2 * (size_t)self->max_batches * self->n), "alloc live sums");
	HIP_TRY(hipMalloc((void **)&self->d_vmax, sizeof(float) * 2 * (size_t)self->max_batches * self->n), "alloc max");
	HIP_TRY(hipMalloc((void **)&self->d_chunk_sum, sizeof(float) * (size_t)(self->max_spectra / 16) * self->n), "alloc chunk sums");
	HIP_TRY(hipMalloc((void **)&self->d_chunk_max, sizeof(float) * (size_t)(self->max_spectra / 16) * self->n), "alloc chunk max");
	HIP_TRY(hipMalloc((void **)&self->d_rise, sizeof(float2) * (kRiseMax + 1)), "alloc rise table");
	HIP_TRY(hipHostMalloc((void **)&self->h_rise, sizeof(float2) * (kRiseMax + 1), hipHostMallocDefault), "alloc pinned rise table");
	HIP_TRY(hipHostMalloc((void **)&self->h_thr, sizeof(double) * (self->n_bins + 1), hipHostMallocDefault), "alloc pinned thr");
	HIP_TRY(hipHostMalloc((void **)&self->h_win, sizeof(float) * self->n, hipHostMallocDefault), "alloc pinned win");

	self->tw_len = build_twiddles(NULL, self->log2n, NULL);
	tw.resize(self->tw_len);
	build_twiddles(tw.data(), self->log2n, self->tw_off);
	HIP_TRY(hipMemcpy(self->d_tw, tw.data(), sizeof(float2) * self->tw_len, hipMemcpyHostToDevice), "upload twiddles");

	/* Initial state (fosphor.c:64-66) */
	fosphor_set_fft_window_default(self);
	fosphor_set_power_range(self, 0, 10);
	self->state = ST_BOOTING;
	return self;

error:
	fosphor_release(self);
	return NULL;
}

extern "C" struct fosphor *fosphor_init(void)
{
	return fosphor_amd_init(NULL);
}

/* ------------------------------------------------------------------------ */
/* Settings                                                                 */
/* ------------------------------------------------------------------------ */

extern "C" void fosphor_set_fft_window_default(struct fosphor *self)
{
	/* periodic Hamming x 1.855, fosphor.c:113-118 */
	for (int i = 0; i < self->n; i++) {
		float ft = (float)self->n;
		float fp = (float)i;
		self->fft_win[i] = (0.54f - 0.46f * cosf((2.0f * 3.141592f * fp) / ft)) * 1.855f;
	}
	self->win_dirty = 1;
}

extern "C" void fosphor_set_fft_window(struct fosphor *self, float *win)
{
	memcpy(self->fft_win, win, sizeof(float) * self->n);		/* fosphor.c:123-128 */
	self->win_dirty = 1;
}

extern "C" void fosphor_set_power_range(struct fosphor *self, int db_ref, int db_per_div)
{
	/* fosphor.c:131-152 */
	int db0 = db_ref - 10 * db_per_div;
	int db1 = db_ref;
	float k = fpm_log10f((float)self->n);
	float offset = -(k + ((float)db0 / 20.0f));
	float scale  = 20.0f / (float)(db1 - db0);

	self->power.db_ref = db_ref;
	self->power.db_per_div = db_per_div;
	self->power.scale = scale;
	self->power.offset = offset;

	/* cl.c:1081-1089, with the bin count a parameter instead of 128 */
	self->histo_scale  = scale * (float)self->n_bins;
	self->histo_offset = offset;
	self->thr_dirty = 1;
}

extern "C" void fosphor_set_frequency_range(struct fosphor *self, double center, double span)
{
	self->frequency.center = center;	/* fosphor.c:154-160 */
	self->frequency.span   = span;
}

/* ------------------------------------------------------------------------ */
/* Launch sequence                                                          */
/* ------------------------------------------------------------------------ */

static void prof_begin(struct fosphor *self, int kind, hipStream_t st)
{
	self->prof_open = 0;
	if (!self->prof || (self->prof == 2 && kind != 0)) return;
	if (self->ev_used + 2 > self->ev_pool.size()) {
		for (int i = 0; i < 2; i++) {
			hipEvent_t e;
			if (hipEventCreateWithFlags(&e, dep_event_flags() & ~hipEventDisableTiming) != hipSuccess) return;
			self->ev_pool.push_back(e);
		}
	}
	if (hipEventRecord(self->ev_pool[self->ev_used], st) != hipSuccess)
		return;				/* this launch goes untimed */
	self->prof_open = 1;
	self->ev_kind.push_back(kind);
}

static void prof_end(struct fosphor *self, hipStream_t st)
{
	if (!self->prof_open) return;
	self->prof_open = 0;
	if (self->ev_used + 2 > self->ev_pool.size() || hipEventRecord(self->ev_pool[self->ev_used + 1], st) != hipSuccess) {
		self->ev_kind.pop_back();	/* drop the half-recorded pair */
		return;
	}
	self->ev_used += 2;
}

static int sync_all(struct fosphor *self)
{
	int rv = 0;
	if (self->stream  && hipStreamSynchronize(self->stream)  != hipSuccess) rv = -EIO;
	for (int i = 1; i < kMaxK1Streams; i++)
		if (self->k1_streams[i] && hipStreamSynchronize(self->k1_streams[i]) != hipSuccess) rv = -EIO;
	if (self->stream2 && hipStreamSynchronize(self->stream2) != hipSuccess) rv = -EIO;
	if (self->stream3 && hipStreamSynchronize(self->stream3) != hipSuccess) rv = -EIO;
	return rv;
}

/* K3s update the persistent state in launch order.  They normally follow each other on one
 * stream; when the stream changes (pipelined process path <-> accumulate/merge path) the new
 * stream waits for the last K3 of the old one. */
static int k3_stream_enter(struct fosphor *self, hipStream_t st)
{
	if (self->last_k3_stream && self->last_k3_stream != st) {
		if (hipEventRecord(self->ev_k3_done, self->last_k3_stream) != hipSuccess ||
		    hipStreamWaitEvent(st, self->ev_k3_done, 0) != hipSuccess)
			return -EIO;
	}
	self->last_k3_stream = st;
	return 0;
}

/* Before anything else than the pipelined 16-bit path writes d_hc / live sums: K3s still
 * reading the two hit-count sets must be done. */
static int drain_h_sets(struct fosphor *self, hipStream_t st)
{
	for (int h = 0; h < 2; h++) {
		if (self->hset_used[h]) {
			if (hipStreamWaitEvent(st, self->ev_h_free[h], 0) != hipSuccess)
				return -EIO;
			self->hset_used[h] = 0;
		}
	}
	return 0;
}

/* Upload lazily-changed tables (cl.c:889-900) and boot fills (cl.c:406-465, 930-934) */
static int prepare(struct fosphor *self)
{
	if (self->win_dirty || self->thr_dirty) {
		/* The tables are read by every K1 still queued -- with relaxed input ordering those of the previous call may be
		 * running on the other FFT streams, which `stream` does not wait for -- and the pinned staging copies (h_win,
		 * h_thr) may still be in flight: drain all FFT streams before either is rewritten. */
		for (int i = 0; i < self->n_k1_streams; i++)
			HIP_TRY(hipStreamSynchronize(self->k1_streams[i]), "drain FFT streams before a table upload");
	}
	if (self->win_dirty) {
		memcpy(self->h_win, self->fft_win, sizeof(float) * self->n);
		HIP_TRY(hipMemcpyAsync(self->d_win, self->h_win, sizeof(float) * self->n, hipMemcpyHostToDevice, self->stream), "upload window");
		self->win_dirty = 0;
	}
	if (self->thr_dirty) {
		build_thresholds(self->h_thr, self->n_bins, self->histo_scale, self->histo_offset);
		HIP_TRY(hipMemcpyAsync(self->d_thr, self->h_thr, sizeof(double) * (self->n_bins + 1), hipMemcpyHostToDevice, self->stream), "upload thresholds");
		self->thr_dirty = 0;
	}
	if (self->state == ST_BOOTING) {
		const float noise_floor = -self->power.offset;
		HIP_TRY(launch_fill((float *)self->d_spectrum, noise_floor, (size_t)4 * self->n, self->stream), "fill spectrum");
		for (int i = 0; i < 2; i++)
			HIP_TRY(launch_fill(self->d_wf_pp[i], noise_floor, (size_t)self->wf_rows * self->n, self->stream), "fill waterfall");
		HIP_TRY(launch_fill(self->d_hist, 0.0f, (size_t)self->n_bins * self->n, self->stream), "fill histogram");
	}
	return 0;
error:
	return -EIO;
}

/* Spectra per K1 wave (N = 1024) / per work-group pass (other lengths).  One tile = one row of live / max
 * partials (8 B per column), so longer tiles mean fewer intermediate bytes: 64 spectra per wave is 0.125 B per
 * sample written and read back, 16 was 0.5.  A tile never straddles a batch (K2 weights whole tiles). */
static int pick_tile(const struct fosphor *self, int total, int batch)
{
	const int v = self->kn_tile;
	if (v >= (self->log2n == 13 ? 8 : 4) && v <= (self->log2n == 16 ? 32 : 128) && !(v & (v - 1)) && batch % v == 0 && total % v == 0)
		return v;
	if (self->log2n == 16 && self->k1h_fused) {
		/* a cluster owns whole tiles: the largest tile that still gives each of the 32 clusters one (at most 32 spectra: the
		 * 9th bits of a tile's bin indices share one dword per column) */
		for (int t = 32; t >= 8; t >>= 1)
			if (batch % t == 0 && total / t >= 32)
				return t;
		return 4;
	}
	if (self->log2n == 13) {
		/* one work-group per CU walks whole tiles; inside a tile the overlapped half of a window is reused from registers,
		 * so long tiles fetch less (tile 64 at 50 % overlap: 65 half-windows for 64 spectra) and leave fewer partial rows;
		 * every CU gets a tile (measured, 16384 spectra per launch: tile 16 / 32 / 64 -> 279 / 288 / 289 GSamples/s) */
		for (int t = 64; t >= 8; t >>= 1)
			if (batch % t == 0 && total / t >= 256)
				return t;
		return 8;		/* (at least 8: the 9th bits of a tile's bin indices go out as one byte per column and eight spectra) */
	}
	if (self->log2n != 10 || self->bins16) {
		/* largest tile that still gives every resident wave (256 CUs x 8) a tile */
		if (total / 16 >= 2048) return 16;
		if (total / 8 >= 2048) return 8;
		return 4;
	}
	/* 1024 waves (one 4-wave work-group per CU) per launch: consecutive launches overlap on alternating
	 * streams, so two of them fill the chip's 512 work-group slots */
	for (int t = 64; t >= 8; t >>= 1)
		if (batch % t == 0 && total / t >= 1024)
			return t;
	return 4;
}

static void fill_k1(struct fosphor *self, K1Params *k1, const void *d_iq, int total, int tile,
                    int wf_pos0, int wf_first, int hop = 0)
{
	const double A = (double)self->histo_scale * 0.150514997831990597606869447362;
	const double C = (double)self->histo_scale * (double)self->histo_offset;
	/* |v_fast - v_pinned| bound, DESIGN.md section 2.3.  Part that does not scale with
	 * |l2|: the float roundings of v itself and of the pinned chain (v < 256: <= 3e-5) plus
	 * histo_scale x (pwr, pwr+offset roundings + the mult by log10(2)/2: <= 8e-7), doubled.
	 * Part proportional to |l2| = |log2 s|: v_log_f32 (<= 1 ulp of l2) and the rounding of
	 * s32 (<= 1.5 ulp of s -> 2.2e-7 in l2, absorbed for |l2| >= 1 ... and by delta0 below
	 * that), through the slope A: kappa = 2 x A x 2^-23. */
	const float delta0 = 6.0e-5f + 2.0e-6f * self->histo_scale;
	const float kappa = (float)(2.0 * A * 1.1920928955078125e-07);

	memset(k1, 0, sizeof(*k1));
	k1->iq = (const float2 *)d_iq;
	k1->hop = hop ? hop : self->n;
	k1->n = self->n; k1->log2n = self->log2n; k1->bins16 = self->bins16;
	for (int q = 0; q < 8; q++) k1->tw_off[q] = self->tw_off[q];
	k1->win = self->d_win;
	k1->tw = self->d_tw;
	k1->thr = self->d_thr;
	k1->bins = self->d_bins;
	k1->partial = self->d_partial;
	k1->wf = self->d_wf_pp[self->wf_cur];
	k1->fft_out = NULL;
	k1->dbg = self->d_dbg;
	k1->total = total;
	k1->tile = tile;
	k1->wf_pos0 = wf_pos0;
	k1->wf_mask = self->wf_rows - 1;
	k1->wf_first = wf_first;
	k1->n_bins = self->n_bins;
	k1->binA = (float)A;
	k1->binC = (float)C;
	k1->amb = 0.5f - delta0;
	k1->kappa = kappa;
	k1->w = 1.0f - self->alpha;		/* display.cl:99 */
	k1->variant = (self->log2n == 10 && !self->bins16) ? self->k1_variant : (self->log2n == 16 ? 4 : 3);
	k1->scratch = self->d_scratch;
	k1->sync = (self->log2n == 16 && self->k1h_fused) ? self->d_k1h_sync : NULL;
	k1->sync_err = self->h_k1h_err;
#ifdef FOSPHOR_AMD_PROBES
	{
		static const int dbg = [] { const char *e = getenv("FOSPHOR_AMD_DBG_K1H"); return e ? atoi(e) : 0; }();
		k1->dbg_k1h = dbg;
	}
#endif
	k1->iq_half = self->iq_half;
	k1->n_cus = self->n_cus;
	{
		const int sc = self->share_cus;
		k1->cus = (self->log2n == 13 && self->overlap && !self->kn_k1w_share_off && sc > 0 && tile > 0 && (total / tile) % sc == 0) ? sc : 0;
		/* ... when there is a tail to share with: the count / merge kernels of the launch before this one are still queued or running
		 * (calls issued back to back).  A launch that finds the chip idle takes all of it.  (The choice depends on timing; the results
		 * do not: the tile loop is grid-stride.  fosphor_amd_share_stats reports how many launches took which form.) */
		if (k1->cus && !(self->tail_set && hipEventQuery(self->ev_tail) == hipErrorNotReady))
			k1->cus = 0;
	}
	if (k1->variant == 1 && (k1->hop & 1))
		k1->variant = 2;		/* 16-byte IQ loads of variant 1 need an even hop */
}

/* A batch longer than 1024 spectra is counted as ONE chunk where that still gives the chip enough work-groups
 * (N/64 slabs per batch) and the (d, e) table covers it: the 16-bit packed counters hold up to 65535 spectra, and K3
 * then reads slab-major 16-bit counts (0.5 B per cell) instead of 32-bit sums of per-chunk slabs. */
static int count_one_chunk(const struct fosphor *self, int batch, int n_batches)
{
	return batch > 1024 && batch <= kRiseMax && (self->n / 64) * n_batches >= 128;
}

/* Sparse K2 -> K3 hand-off (row masks + hot flags) for the large state of N = 65536 (128 MiB, one batch = one frame: +10 % for
 * the path).  At N = 1024 / 8192 it measured slower (K3 there is bound by its dependent chain over the batches of a launch,
 * not by the rows it touches: -1 % / -2.5 %) and is not offered.  FOSPHOR_AMD_ROWMASK=0 selects the dense form at N = 65536
 * (the tests run both). */
static int use_rowmask(const struct fosphor *self)
{
	return self->log2n == 16 && !self->kn_rowmask_off;
}

static int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

/* Batches per sub-launch ("piece") of a device-resident call of n_batches batches: about sub_samples samples each, pieces of equal
 * size.  N = 8192 with the streams on: pieces of whole `unit`s of batches where the call allows it, so that every piece's tiles are a
 * multiple of kK1wShareCus and the FFT launch can leave CUs to the count / merge kernels (tiles are 64 spectra or the batch:
 * pick_tile).  Pure host arithmetic, exported for the CPU test suite (include/fosphor_amd.h). */
extern "C" int fosphor_amd_plan_piece_batches(int log2n, int overlap, int n_batches, int batch, long long sub_samples)
{
	if (n_batches < 1 || batch < 1 || log2n < 1 || log2n > 30 || sub_samples < 1)
		return -EINVAL;
	const long long per_batch = (long long)batch << log2n;
	const int cap = (int)(sub_samples / per_batch < 1 ? 1 : (sub_samples / per_batch > n_batches ? n_batches : sub_samples / per_batch));
	const int n_sub = (n_batches + cap - 1) / cap;
	int sub_b = (n_batches + n_sub - 1) / n_sub;
	if (log2n == 13 && overlap && n_sub > 1) {
		const int tpb = batch >= 64 ? batch / 64 : 1;
		const int unit = kK1wShareCus / gcd_int(kK1wShareCus, tpb);
		if (n_batches % unit == 0 && cap >= unit)
			sub_b = cap / unit * unit;
	}
	return sub_b;
}

/* K2 (+K2b) for n_batches batches of `batch` spectra whose bin indices / tile partials are in
 * d_bins / d_partial; results land in slot `slot0`.. of hc / live_sum / vmax. */
static int run_count(struct fosphor *self, int n_batches, int batch, int tile, int slot0,
                     int t_offset, int weight_batch, hipStream_t st, int use16 = 0, int hset = 0)
{
	K2Params k2; K2bParams k2b;
	const int one_chunk = use16 && count_one_chunk(self, batch, n_batches);
	const int chunk = (batch <= 1024 || one_chunk) ? batch : gcd_int(batch, 1024);
	const int cpb = batch / chunk;
	const size_t cells = (size_t)self->n_bins * self->n;

	/* Batches of several 1024-spectrum chunks (a batch longer than the reference's cap, or the time shard of
	 * a display frame): K2 leaves per-chunk packed 16-bit slabs in d_slab16 (no zeroing, no global atomics)
	 * and k2c_sum adds each batch's slabs into its 32-bit array.  Needs whole chunks and room for the slabs. */
	const int sum16 = (!use16 || batch > 1024) && chunk == 1024 && cpb > 1 && self->d_slab16 &&
	                  n_batches * cpb <= self->slab_chunks && !self->kn_no_sum16;

	memset(&k2, 0, sizeof(k2));
	k2.bins = self->d_bins; k2.partial = self->d_partial;
	k2.hc = self->d_hc + (size_t)slot0 * cells;
	k2.hc16 = (use16 && (batch <= 1024 || one_chunk) && self->rise_ok(batch))
	          ? (uint16_t *)self->d_hc + (size_t)hset * self->max_batches * cells : NULL;
	if (k2.hc16 && use_rowmask(self)) {
		k2.rowmask = self->d_rowmask + (size_t)hset * self->max_batches * (self->n / 64) * self->mask_words;
		k2.mask_words = self->mask_words;
		k2.mask_stride = self->max_batches;
	}
	if (sum16)
		k2.hc16 = self->d_slab16, k2.rowmask = NULL;
	const int lslot = slot0 + hset * self->max_batches;	/* live-sum / max slot */
	k2.n = self->n; k2.bins16 = self->bins16 && self->log2n != 13; k2.bins8p1 = (self->log2n == 13);
	k2.bins9 = (self->log2n == 16); k2.total = n_batches * batch;
	k2.batch = batch; k2.chunk = chunk; k2.tile = tile; k2.n_bins = self->n_bins;
	k2.w = 1.0f - self->alpha;
	k2.log2_w = (float)log2((double)(1.0f - self->alpha));
	k2.t_offset = t_offset; k2.weight_batch = weight_batch;
#ifdef FOSPHOR_AMD_PROBES
	{
		static const int dbg_same = getenv("FOSPHOR_AMD_DBG_SAME") != NULL;
		k2.dbg_same = dbg_same;
	}
#endif
	if (cpb == 1) {
		k2.chunk_sum = self->d_live_sum + (size_t)lslot * self->n;
		k2.chunk_max = self->d_vmax + (size_t)lslot * self->n;
	} else {
		k2.chunk_sum = self->d_chunk_sum;
		k2.chunk_max = self->d_chunk_max;
		if (!sum16)
			HIP_TRY(hipMemsetAsync(k2.hc, 0, sizeof(uint32_t) * cells * n_batches, st), "zero hit counts");
	}
	prof_begin(self, 1, st);
	HIP_TRY(launch_k2(k2, n_batches * cpb, st), "launch count");
	self->k2_stream_last = st;
	if (cpb > 1) {
		memset(&k2b, 0, sizeof(k2b));
		k2b.chunk_sum = self->d_chunk_sum; k2b.chunk_max = self->d_chunk_max;
		k2b.live_sum = self->d_live_sum + (size_t)lslot * self->n;
		k2b.vmax = self->d_vmax + (size_t)lslot * self->n;
		k2b.n_batches = n_batches; k2b.cpb = cpb; k2b.n = self->n;
		if (sum16) {
			k2b.hc16 = k2.hc16; k2b.hc = k2.hc; k2b.n_bins = self->n_bins;
			HIP_TRY(launch_k2c(k2b, st), "launch chunk sum");
		} else {
			HIP_TRY(launch_k2b(k2b, st), "launch chunk reduce");
		}
	}
	prof_end(self, st);
	return 0;
error:
	return -EIO;
}

/* (d, e) of display.cl:241-245 for every possible hit count of a batch.  Same float
 * expressions as the kernel source, powf from the host libm (the oracle's binding). */
static int ensure_rise_table(struct fosphor *self, int batch, hipStream_t st)
{
	if (batch > kRiseMax)
		return 0;
	if (self->rise_batch == batch && self->rise_t0r == self->t0r && self->rise_t0d == self->t0d)
		return 1;
	if (sync_all(self))				/* h_rise may be in flight */
		return -1;
	for (int hc = 0; hc <= batch; hc++) {
		const float a = (float)hc / (float)batch;
		const float b = a * (1.0f / self->t0r);
		const float c = b + (1.0f / self->t0d);
		const float d = b * (1.0f / c);
		const float e = powf(1.0f - c, (float)batch);
		self->h_rise[hc] = make_float2(d, e);
	}
	if (hipMemcpyAsync(self->d_rise, self->h_rise, sizeof(float2) * (batch + 1), hipMemcpyHostToDevice, st) != hipSuccess)
		return -1;
	self->rise_batch = batch; self->rise_t0r = self->t0r; self->rise_t0d = self->t0d;
	return 1;
}

static int run_merge(struct fosphor *self, int n_batches, int batch, int slot0, hipStream_t st, int use16 = 0, int hset = 0,
                     int cell_begin = 0, int cell_end = 0)
{
	K3Params k3;
	const size_t cells = (size_t)self->n_bins * self->n;
	const int lslot = slot0 + hset * self->max_batches;
	const int have_table = ensure_rise_table(self, batch, st);
	if (have_table < 0 || k3_stream_enter(self, st))
		return -EIO;
	memset(&k3, 0, sizeof(k3));
	k3.rise = have_table ? self->d_rise : NULL;
	k3.live_decay = powf(1.0f - self->alpha, (float)batch);	/* display.cl:210 */
	k3.hc = self->d_hc + (size_t)slot0 * cells;
	const int one_chunk = use16 && count_one_chunk(self, batch, n_batches);
	k3.hc16 = (use16 && (batch <= 1024 || one_chunk) && have_table)
	          ? (const uint16_t *)self->d_hc + (size_t)hset * self->max_batches * cells : NULL;
	k3.hc_export = self->d_hc_export;
	if (k3.hc16) {
		/* the uint32 view of the last batch is made when somebody asks for it (fosphor_amd_get_buffers) */
		const size_t per_batch = (size_t)(self->n / 64) * self->mask_words;
		k3.hc_export = NULL;
		self->export_src = k3.hc16 + (size_t)(n_batches - 1) * cells;
		self->export_mask = NULL;
		self->export_stream = self->k2_stream_last ? self->k2_stream_last : st;
		if (use_rowmask(self)) {
			k3.rowmask = self->d_rowmask + (size_t)hset * self->max_batches * per_batch;
			k3.mask_words = self->mask_words;
			k3.mask_stride = self->max_batches;
			self->export_mask = k3.rowmask + (n_batches - 1);
			k3.hot = self->d_hot;
			k3.rowlist = self->d_rowlist;
			k3.rowlist_cnt = self->rowlist_flip ? 1 + self->n_bins * (self->n / 64) : 0;
			self->rowlist_flip ^= 1;
			k3.hot_all = !self->hot_valid;
			self->hot_valid = 1;
		} else {
			self->hot_valid = 0;
		}
	} else {
		self->export_src = NULL;
		self->hot_valid = 0;
	}
	k3.live_sum = self->d_live_sum + (size_t)lslot * self->n;
	k3.vmax = self->d_vmax + (size_t)lslot * self->n;
	k3.hist = self->d_hist; k3.spectrum = self->d_spectrum;
	k3.n_batches = n_batches; k3.batch = batch; k3.n_bins = self->n_bins; k3.n = self->n;
	k3.t0r = self->t0r; k3.t0d = self->t0d; k3.alpha = self->alpha;
	k3.cell_begin = cell_begin; k3.cell_end = cell_end;
#ifdef FOSPHOR_AMD_PROBES
	{
		static const int dbg_same = getenv("FOSPHOR_AMD_DBG_SAME") != NULL;
		k3.dbg_same = dbg_same;
	}
#endif
	prof_begin(self, 2, st);
	HIP_TRY(launch_k3(k3, st), "launch merge");
	prof_end(self, st);
	return 0;
error:
	return -EIO;
}

/* the other FFT streams see what the caller (and prepare()) queued on `stream` ... */
static int k1_streams_fork(struct fosphor *self)
{
	if (hipEventRecord(self->ev_in, self->stream) != hipSuccess)
		return -EIO;
	for (int i = 1; i < self->n_k1_streams; i++)
		if (hipStreamWaitEvent(self->k1_streams[i], self->ev_in, 0) != hipSuccess)
			return -EIO;
	return 0;
}

/* ... and what the caller queues on `stream` next follows every K1 queued on them */
static int k1_streams_join(struct fosphor *self)
{
	for (int i = 1; i < self->n_k1_streams; i++)
		if (hipEventRecord(self->ev_k1s_done[i], self->k1_streams[i]) != hipSuccess ||
		    hipStreamWaitEvent(self->stream, self->ev_k1s_done[i], 0) != hipSuccess)
			return -EIO;
	return 0;
}

/* Waterfall ring ownership between K1s that may run on different streams: a K1 that stores rows into ring b
 * follows the previous K1 that did. */
static int wf_enter(struct fosphor *self, hipStream_t ks)
{
	const int b = self->wf_cur;
	if (self->wf_used[b] && self->wf_stream[b] != ks && hipStreamWaitEvent(ks, self->ev_wf[b], 0) != hipSuccess)
		return -EIO;
	return 0;
}

static int wf_leave(struct fosphor *self, hipStream_t ks)
{
	const int b = self->wf_cur;
	if (hipEventRecord(self->ev_wf[b], ks) != hipSuccess)
		return -EIO;
	self->wf_used[b] = 1;
	self->wf_stream[b] = ks;
	return 0;
}

/* n_batches consecutive batches of `batch` spectra.  device_call: the samples are the caller's device buffer
 * (fosphor_amd_process_device*), which may be cut into sub-launches whose K1s alternate between two streams. */
static int run(struct fosphor *self, const void *d_iq, int n_batches, int batch, int hop = 0, int device_call = 0)
{
	const int total = n_batches * batch;
	const size_t sample_bytes = self->iq_half ? 4 : sizeof(float2);
	const int hop_samples = hop ? hop : self->n;
	hipStream_t st2 = self->overlap ? self->stream2 : self->stream;
	/* third stream: only the 16-bit count path has a second hit-count set */
	const int three = self->overlap && self->pipe3 && batch <= 1024 && self->rise_ok(batch);
	hipStream_t st3 = three ? self->stream3 : st2;
	const int did_prep = self->win_dirty || self->thr_dirty || self->state == ST_BOOTING;
	const int wf_first_global = total > self->wf_rows ? total - self->wf_rows : 0;
	int sub_b, n_sub, use_alt, used_alt = 0;
	/* measurement only, probe builds (results are wrong): FOSPHOR_AMD_DBG_SKIP bit 0 = no K1, bit 1 = no count / merge, bit 2 = no K3,
	 * bit 3 = no K2; FOSPHOR_AMD_DBG_NOWAIT: an FFT launch does not wait for the count kernel that still reads its intermediate set */
#ifdef FOSPHOR_AMD_PROBES
	static const int dbg_skip = [] { const char *e = getenv("FOSPHOR_AMD_DBG_SKIP"); return e ? atoi(e) : 0; }();
	static const int dbg_nowait = getenv("FOSPHOR_AMD_DBG_NOWAIT") != NULL;
#else
	constexpr int dbg_skip = 0, dbg_nowait = 0;
#endif

	if (prepare(self))
		return -EIO;

	/* Sub-launches.  A call is cut into pieces of about sub_samples samples (64 reference batches): the
	 * bin-index / partial intermediates of a piece stay small enough to be consumed by K2 out of the Infinity
	 * Cache, and the pipeline below overlaps K2/K3 of piece j with K1 of piece j+1 INSIDE one call.
	 *   K1 (j)   on `stream` and the other FFT streams in rotation: K1 of piece j+1 is dispatched while K1 of piece j
	 *            drains, so no CU idles between them (kernel tail, dispatch gap and prologue overlap);
	 *   K2 (j)   on stream2 once K1 (j) has finished; K3 (j) follows it there, so the persistent state sees the
	 *            batches in order.
	 * The intermediates rotate among kSets sets: K1 may reuse a set once the K2 that read it has finished. */
	sub_b = fosphor_amd_plan_piece_batches(self->log2n, self->overlap, n_batches, batch, self->sub_samples);
	n_sub = (n_batches + sub_b - 1) / sub_b;
	use_alt = self->overlap && self->alt && device_call && (n_sub > 1 || self->relaxed);
	if (self->log2n == 16 && self->k1h_fused)
		use_alt = 0;		/* one fused FFT kernel at a time: its clusters own the counters and the intermediate */

	/* A call that stores every row of the ring does so in the other ring: its K1s then owe nothing to the
	 * row stores of the calls before it.  Otherwise the untouched rows must survive: same ring. */
	if (use_alt && total >= self->wf_rows)
		self->wf_cur ^= 1;
	if (use_alt && (did_prep || !self->relaxed)) {
		if (k1_streams_fork(self))
			return -EIO;
	}

	for (int b0 = 0; b0 < n_batches; b0 += sub_b) {
		const int nb = (n_batches - b0 < sub_b) ? n_batches - b0 : sub_b;
		const int t0 = b0 * batch, sub_total = nb * batch;
		const int tile = pick_tile(self, sub_total, batch);
		hipStream_t ks = self->stream;
		K1Params k1;
		int set, hset = 0, wf_first, stores_rows;

		if (use_alt) {
			ks = self->k1_streams[self->k1_seq++ % self->n_k1_streams];
			used_alt = 1;
		}
		set = self->pp;
		self->pp = (self->pp + 1) % self->n_sets;
		self->d_bins = self->d_bins_pp[set];
		self->d_partial = self->d_partial_pp[set];
		if (self->overlap && self->set_used[set] && !dbg_nowait)
			HIP_TRY(hipStreamWaitEvent(ks, self->ev_set_free[set], 0), "wait for intermediate set");
		wf_first = wf_first_global - t0;
		if (wf_first < 0) wf_first = 0;
		stores_rows = wf_first < sub_total;
		if (!stores_rows) wf_first = sub_total;
		fill_k1(self, &k1, (const char *)d_iq + (size_t)t0 * hop_samples * sample_bytes, sub_total, tile,
		        (self->wf_pos + t0) & (self->wf_rows - 1), wf_first, hop);
		if (stores_rows && wf_enter(self, ks))
			return -EIO;
		prof_begin(self, 0, ks);
		if (!(dbg_skip & 1))
			HIP_TRY(launch_k1(k1, ks), "launch fft_bin");
		prof_end(self, ks);
		if (self->log2n == 13) {
			if (k1.cus) self->k1w_shared++; else self->k1w_full++;
		}
		if (stores_rows && wf_leave(self, ks))
			return -EIO;

		if (self->overlap) {
			HIP_TRY(hipEventRecord(self->ev_k1_done[set], ks), "record K1 done");
			HIP_TRY(hipStreamWaitEvent(st2, self->ev_k1_done[set], 0), "K2 waits for K1");
		}
		if (dbg_skip & 2) {
			if (self->overlap) {
				HIP_TRY(hipEventRecord(self->ev_set_free[set], st2), "record set free");
				self->set_used[set] = 1;
			}
			continue;
		}
		if (three) {
			hset = self->hset;
			self->hset ^= 1;
			if (self->hset_used[hset])
				HIP_TRY(hipStreamWaitEvent(st2, self->ev_h_free[hset], 0), "wait for hit-count set");
		} else if (drain_h_sets(self, st2)) {
			return -EIO;
		}
		if (!(dbg_skip & 8) && run_count(self, nb, batch, tile, 0, 0, batch, st2, 1, hset))
			return -EIO;
		if (self->overlap) {
			HIP_TRY(hipEventRecord(self->ev_set_free[set], st2), "record set free");
			self->set_used[set] = 1;
		}
		if (three) {
			HIP_TRY(hipEventRecord(self->ev_k2_done[hset], st2), "record K2 done");
			HIP_TRY(hipStreamWaitEvent(st3, self->ev_k2_done[hset], 0), "K3 waits for K2");
		}
		if (!(dbg_skip & 4) && run_merge(self, nb, batch, 0, st3, 1, hset))
			return -EIO;
		if (self->overlap && self->log2n == 13) {
			HIP_TRY(hipEventRecord(self->ev_tail, st3), "record tail");
			self->tail_set = 1;
		}
		if (three) {
			HIP_TRY(hipEventRecord(self->ev_h_free[hset], st3), "record hit-count set free");
			self->hset_used[hset] = 1;
		}
		self->last_batches = nb;
	}
	if (used_alt && !self->relaxed) {
		/* what the caller queues on `stream` next (e.g. refilling the sample buffer) follows every K1 */
		if (k1_streams_join(self))
			return -EIO;
	}
	self->last_hc16 = (batch <= 1024 || count_one_chunk(self, batch, self->last_batches));

	self->wf_pos = (self->wf_pos + total) & (self->wf_rows - 1);	/* cl.c:954 */
	self->last_slot0 = 0;
	self->state = ST_PENDING;
	return 0;
error:
	return -EIO;
}

extern "C" int fosphor_amd_process_device(struct fosphor *self, const void *d_samples, int n_batches, int batch)
{
	if (!self || !d_samples || n_batches < 1 || batch < 16 || (batch & 15))
		return -EINVAL;
	if ((long long)n_batches * batch > self->max_spectra || n_batches > self->max_batches)
		return -EINVAL;
	return run(self, d_samples, n_batches, batch, 0, 1);
}

/* overlap_cc (lib/overlap_cc_impl.cc:48-79) emits wlen-sample windows whose starts advance
 * wlen/overlap input samples, i.e. it materialises an overlap-times larger stream for the sink.
 * Here the same windows are read straight from the unexpanded stream: spectrum t starts at
 * sample t * N / overlap, so the unique HBM read per FFT'd sample drops to 8/overlap bytes. */
extern "C" int fosphor_amd_process_device_overlap(struct fosphor *self, const void *d_samples,
                                                  int n_batches, int batch, int overlap)
{
	if (!self || !d_samples || n_batches < 1 || batch < 16 || (batch & 15))
		return -EINVAL;
	if (overlap < 1 || overlap > self->n || (self->n % overlap))
		return -EINVAL;
	if ((long long)n_batches * batch > self->max_spectra || n_batches > self->max_batches)
		return -EINVAL;
	return run(self, d_samples, n_batches, batch, self->n / overlap, 1);
}

extern "C" int fosphor_process(struct fosphor *self, void *samples, int len)
{
	int k;
	const size_t sample_bytes = self->iq_half ? 4 : sizeof(float2);

	/* cl.c:882-886 */
	if (len <= 0 || (len & ((16 * self->n) - 1)))
		return -EINVAL;
	if ((long long)len > (long long)self->n * 1024)
		return -EINVAL;
	if (len / self->n > self->max_spectra)
		return -EINVAL;

	/* cl.c:903-910 enqueues a non-blocking write straight from the caller's memory, which the
	 * sink reuses immediately (base_sink_c_impl.cc:168-174: a latent race).  Here the samples
	 * are copied into a pinned ring slot before returning, and the slot is recycled only when
	 * its H2D copy has completed. */
	/* (uploads queued by fosphor_amd_upload_pinned and not yet processed come first: the sample stream is applied in order, and a
	 * pending upload may own the staging slot this call is about to fill) */
	while (self->pend_n) {
		const int rv = fosphor_amd_process_uploaded(self, NULL);
		if (rv)
			return rv;
	}
	k = self->stage_idx;
	/* (each piece behind its own check: fosphor_amd_process_pinned shares d_stage / stage_free) */
	if (!self->h_stage[k])
		HIP_TRY(hipHostMalloc((void **)&self->h_stage[k], sample_bytes * self->stage_samples, hipHostMallocDefault), "alloc pinned staging");
	if (!self->d_stage[k]) {
		HIP_TRY(hipMalloc((void **)&self->d_stage[k], sample_bytes * self->stage_samples), "alloc device staging");
		self->d_stage_cap[k] = self->stage_samples;
	}
	if (!self->stage_free[k])
		HIP_TRY(hipEventCreateWithFlags(&self->stage_free[k], hipEventDisableTiming), "create staging event");
	else
		HIP_TRY(hipEventSynchronize(self->stage_free[k]), "wait staging slot");
	memcpy(self->h_stage[k], samples, sample_bytes * (size_t)len);
	HIP_TRY(hipMemcpyAsync(self->d_stage[k], self->h_stage[k], sample_bytes * (size_t)len, hipMemcpyHostToDevice, self->stream), "H2D samples");
	{
		int rv = run(self, self->d_stage[k], 1, len / self->n);
		/* the slot is free again once everything queued so far (copy + kernels reading
		 * d_stage[k]) has finished */
		if (hipEventRecord(self->stage_free[k], self->stream) != hipSuccess && !rv)
			rv = -EIO;
		self->stage_used[k] = 1;		/* a later fosphor_amd_upload_pinned into this slot (its own stream) waits for stage_free[k] */
		self->stage_idx ^= 1;
		return rv;
	}
error:
	return -EIO;
}

/* The upload of fosphor_amd_process_pinned and its kernels as two calls: a host that queues the NEXT upload before it waits for the
 * kernels of this one (the streaming sink: its per-frame synchronisation point drains the kernel streams, not the upload stream)
 * keeps the link busy across frames.  At most two uploads can be pending (two staging buffers): -EBUSY beyond. */
extern "C" int fosphor_amd_upload_pinned(struct fosphor *self, const void *samples, int len)
{
	int k;
	const size_t sample_bytes = self->iq_half ? 4 : sizeof(float2);

	/* cl.c:882-886 for one batch; beyond it a whole number of 1024-spectrum batches in one call (applied one after the other like so
	 * many calls, one upload and one set of launches: what the host spends per call -- ~250 us -- is then spent per 64 MiB, not per 8) */
	const long long one = (long long)self->n * 1024;
	if (len <= 0 || (len & ((16 * self->n) - 1)) || len / self->n > self->max_spectra || ((long long)len > one && (long long)len % one))
		return -EINVAL;
	if (((long long)len > one ? (int)((long long)len / one) : 1) > self->max_batches)
		return -EINVAL;
	if (self->pend_n == 2)
		return -EBUSY;

	k = self->stage_idx;
	if (self->d_stage[k] && self->d_stage_cap[k] < (size_t)len) {		/* grown on demand (behind everything that reads the old one) */
		/* (this instance's streams only: the caller's unrelated streams are not stalled) */
		if (sync_all(self) || (self->copy_stream && hipStreamSynchronize(self->copy_stream) != hipSuccess))
			return -EIO;
		(void)hipFree(self->d_stage[k]);
		self->d_stage[k] = NULL;
	}
	if (!self->d_stage[k]) {
		const size_t cap = (size_t)len > self->stage_samples ? (size_t)len : self->stage_samples;
		HIP_TRY(hipMalloc((void **)&self->d_stage[k], sample_bytes * cap), "alloc device staging");
		self->d_stage_cap[k] = cap;
	}
	if (!self->stage_free[k])
		HIP_TRY(hipEventCreateWithFlags(&self->stage_free[k], hipEventDisableTiming), "create staging event");
	if (!self->upload_done)
		HIP_TRY(hipEventCreateWithFlags(&self->upload_done, hipEventDisableTiming), "create upload event");
	if (!self->copy_stream)
		HIP_TRY(hipStreamCreateWithFlags(&self->copy_stream, hipStreamNonBlocking), "create upload stream");
	if (!self->upload_slot[k])
		HIP_TRY(hipEventCreateWithFlags(&self->upload_slot[k], hipEventDisableTiming), "create upload event");
	/* The upload runs on its own stream, beside the kernels of the batch before it (on one stream a batch cost its 150 us of DMA PLUS
	 * its kernels: 3.9 GSamples/s where the link does 7).  d_stage[k] was last read by the FFT kernel of the call before the last:
	 * the copy waits for the event recorded behind that call; the instance's stream waits for the copy. */
	if (self->stage_used[k])
		HIP_TRY(hipStreamWaitEvent(self->copy_stream, self->stage_free[k], 0), "upload waits for the staging slot");
	HIP_TRY(hipMemcpyAsync(self->d_stage[k], samples, sample_bytes * (size_t)len, hipMemcpyHostToDevice, self->copy_stream), "H2D samples (pinned)");
	HIP_TRY(hipEventRecord(self->upload_slot[k], self->copy_stream), "record upload");
	HIP_TRY(hipEventRecord(self->upload_done, self->copy_stream), "record upload");
	self->pend_slot[(self->pend_head + self->pend_n) & 1] = k;
	self->pend_len[(self->pend_head + self->pend_n) & 1] = len;
	self->pend_n++;
	self->stage_idx ^= 1;
	return 0;
error:
	return -EIO;
}

extern "C" int fosphor_amd_pending_uploads(struct fosphor *self)
{
	return self ? self->pend_n : 0;
}

/* The kernels for the oldest pending upload; *len_out = its samples.  -EINVAL when nothing is pending. */
extern "C" int fosphor_amd_process_uploaded(struct fosphor *self, int *len_out)
{
	if (!self || !self->pend_n)
		return -EINVAL;
	const int k = self->pend_slot[self->pend_head], len = self->pend_len[self->pend_head];
	const long long one = (long long)self->n * 1024;
	const int n_batches = (long long)len > one ? (int)((long long)len / one) : 1;
	int rv;
	self->pend_head ^= 1;
	self->pend_n--;
	if (len_out)
		*len_out = len;
	HIP_TRY(hipStreamWaitEvent(self->stream, self->upload_slot[k], 0), "kernels wait for the upload");
	rv = run(self, self->d_stage[k], n_batches, len / self->n / n_batches);
	if (hipEventRecord(self->stage_free[k], self->stream) != hipSuccess && !rv)
		rv = -EIO;
	self->stage_used[k] = 1;
	return rv;
error:
	return -EIO;
}

extern "C" int fosphor_amd_process_pinned(struct fosphor *self, const void *samples, int len)
{
	int rv;
	/* (uploads queued by fosphor_amd_upload_pinned and not yet processed come first: the stream is applied in order) */
	while (self->pend_n)
		if ((rv = fosphor_amd_process_uploaded(self, NULL)) != 0)
			return rv;
	if ((rv = fosphor_amd_upload_pinned(self, samples, len)) != 0)
		return rv;
	return fosphor_amd_process_uploaded(self, NULL);
}

extern "C" int fosphor_amd_wait_upload(struct fosphor *self)
{
	if (!self || !self->upload_done)
		return 0;
	return hipEventSynchronize(self->upload_done) == hipSuccess ? 0 : -EIO;
}

extern "C" int fosphor_amd_finish(struct fosphor *self)
{
	if (self->state == ST_READY)
		return 0;				/* cl.c:977-979 */
	if (self->state == ST_BOOTING) {		/* cl.c:981-995: finish the boot */
		if (prepare(self))
			return -EIO;
	}
	if (sync_all(self))
		return -EIO;
	if (self->h_k1h_err && self->h_k1h_err[0]) {
		fprintf(stderr, "[fosphor_amd] fused 65536-point FFT: a cluster wait timed out (code 0x%x); "
		        "results are invalid\n", self->h_k1h_err[0]);
		self->h_k1h_err[0] = 0;
		return -EIO;
	}
	self->state = ST_READY;
	return 1;
}

extern "C" void fosphor_draw(struct fosphor *self, struct fosphor_render *render)
{
	(void)fosphor_amd_finish(self);			/* fosphor.c:100-101 */
	if (render)
		render->_wf_pos = self->wf_pos;		/* fosphor.c:103 */
}

/* ------------------------------------------------------------------------ */
/* Buffers                                                                  */
/* ------------------------------------------------------------------------ */

static int get_buffers(struct fosphor *self, struct fosphor_amd_buffers *out, int want_hitcount)
{
	if (!self || !out)
		return -EINVAL;
	if (want_hitcount && self->last_hc16 && self->export_src) {
		/* queued on the stream of the K2 that wrote the slabs (stream order = behind it); the view is complete when
		 * this call returns */
		hipStream_t st = self->export_stream ? self->export_stream : self->stream;
		if (launch_export_hc16(self->export_src, self->export_mask, self->mask_words, self->max_batches, self->d_hc_export, self->n_bins, self->n, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess)
			return -EIO;
		self->export_src = NULL;
	}
	out->d_waterfall = self->d_wf_pp[self->wf_cur];
	out->d_histogram = self->d_hist;
	out->d_spectrum  = (float *)self->d_spectrum;
	out->d_hitcount  = !want_hitcount ? NULL : self->last_hc16 ? self->d_hc_export
	                   : self->d_hc + (size_t)(self->last_slot0 + (self->last_batches > 0 ? self->last_batches - 1 : 0)) * self->n_bins * self->n;
	out->waterfall_pos = self->wf_pos;
	out->fft_len = self->n; out->n_bins = self->n_bins; out->wf_rows = self->wf_rows;
	out->histo_scale = self->histo_scale; out->histo_offset = self->histo_offset;
	return 0;
}

extern "C" int fosphor_amd_get_buffers(struct fosphor *self, struct fosphor_amd_buffers *out)
{
	return get_buffers(self, out, 1);
}

/* The same without the hit-count view (d_hitcount = NULL): nothing is launched, nothing is waited for. */
extern "C" int fosphor_amd_get_buffers_nohc(struct fosphor *self, struct fosphor_amd_buffers *out)
{
	return get_buffers(self, out, 0);
}

extern "C" int fosphor_amd_read(struct fosphor *self, int which, void *host, uint64_t bytes)
{
	struct fosphor_amd_buffers b;
	const void *src; uint64_t want;
	int rv;

	if (!self || !host)
		return -EINVAL;
	rv = fosphor_amd_finish(self);
	if (rv < 0)
		return rv;
	rv = get_buffers(self, &b, which == 3);
	if (rv < 0)
		return rv;
	switch (which) {
	case 0: src = b.d_waterfall; want = sizeof(float) * (uint64_t)self->wf_rows * self->n; break;
	case 1: src = b.d_histogram; want = sizeof(float) * (uint64_t)self->n_bins * self->n; break;
	case 2: src = b.d_spectrum;  want = sizeof(float) * 4 * self->n; break;
	case 3: src = b.d_hitcount;  want = sizeof(uint32_t) * (uint64_t)self->n_bins * self->n; break;
	default: return -EINVAL;
	}
	if (bytes != want)
		return -EINVAL;
	if (hipMemcpy(host, src, want, hipMemcpyDeviceToHost) != hipSuccess)
		return -EIO;
	return 0;
}

/* ------------------------------------------------------------------------ */
/* Test hooks                                                               */
/* ------------------------------------------------------------------------ */

extern "C" int fosphor_amd_fft(struct fosphor *self, const void *d_in, void *d_out, int n_spectra)
{
	K1Params k1;
	int saved_state;
	if (!self || !d_in || !d_out || n_spectra < 4 || (n_spectra & 3) || n_spectra > self->max_spectra)
		return -EINVAL;
	if (sync_all(self))				/* scratch sets may still be read by a queued K2 */
		return -EIO;
	saved_state = self->state;
	self->state = ST_READY;			/* no boot fills for a pure FFT */
	if (prepare(self)) { self->state = saved_state; return -EIO; }
	self->state = saved_state;
	/* rows are not stored (wf_first = total); bins/partials land in scratch */
	/* (N = 8192: tiles of at least 8 spectra -- the 9th bits of the bin indices go out per eight spectra) */
	if (self->log2n == 13 && (n_spectra & 7))
		return -EINVAL;
	fill_k1(self, &k1, d_in, n_spectra, self->log2n == 13 ? 8 : 4, 0, n_spectra);
	k1.fft_out = (float2 *)d_out;
	if (launch_k1(k1, self->stream) != hipSuccess)
		return -EIO;
	return hipStreamSynchronize(self->stream) == hipSuccess ? 0 : -EIO;
}

extern "C" int fosphor_amd_bin(struct fosphor *self, const void *d_fft, void *d_bin, void *d_pwr, int n)
{
	K1Params k1;
	int saved_state, force = 0;
	const char *e = getenv("FOSPHOR_AMD_FORCE_EXACT_BIN");
	if (!self || !d_fft || !d_bin || !d_pwr || n < 1)
		return -EINVAL;
	if (e && *e == '1') force = 1;
	if (sync_all(self))
		return -EIO;
	saved_state = self->state;
	self->state = ST_READY;
	if (prepare(self)) { self->state = saved_state; return -EIO; }
	self->state = saved_state;
	fill_k1(self, &k1, NULL, 0, 4, 0, 0);
	if (launch_bin_hook((const float2 *)d_fft, (uint8_t *)d_bin, (float *)d_pwr, n, k1, force, self->stream) != hipSuccess)
		return -EIO;
	return hipStreamSynchronize(self->stream) == hipSuccess ? 0 : -EIO;
}

/* ------------------------------------------------------------------------ */
/* Multi-GPU split                                                          */
/* ------------------------------------------------------------------------ */

static int accumulate(struct fosphor *self, const void *d_samples, int n_local, int t_offset, int total_batch, int hop)
{
	K1Params k1;
	int tile, wf_first, set;
	hipStream_t st2;

	if (!self || !d_samples || n_local < 16 || (n_local & 15) || (t_offset & 15) ||
	    t_offset < 0 || t_offset + n_local > total_batch || n_local > self->max_spectra)
		return -EINVAL;
	const int did_prep = self->win_dirty || self->thr_dirty || self->state == ST_BOOTING;
	if (prepare(self))
		return -EIO;

	st2 = self->overlap ? self->stream2 : self->stream;
	{
		/* A shard of several whole 1024-spectrum chunks runs like a device-resident call: sub-launches of
		 * sub_samples samples, K1s alternating between two streams, K2 of piece j beside K1 of piece j + 1.
		 * Each K2 leaves its chunks' packed 16-bit count slabs and float partials at the chunks' places;
		 * one k2c_sum at the end adds all of them into the 32-bit slot that is exchanged. */
		const int cpb = n_local / 1024;
		const size_t cells = (size_t)self->n_bins * self->n;
		const int chunked = (n_local % 1024) == 0 && cpb > 1 && self->d_slab16 && cpb <= self->slab_chunks &&
		                    !self->kn_no_sum16;
		long long per_chunk = 1024LL * self->n;
		int sub_c = (int)(self->sub_samples / per_chunk);
		if (sub_c < 1) sub_c = 1;
		if (chunked && cpb > sub_c) {
			const size_t sample_bytes = self->iq_half ? 4 : sizeof(float2);
			/* K2 counts G consecutive 1024-spectrum chunks per work-group (16-bit counters hold 65535 spectra): 1 / G of the
			 * slab traffic, as long as enough work-groups are left to keep the bin-index reads in flight */
			int G = 1;
			{
				/* measured, N = 1024, 256 batches per frame: G = 1 / 2 / 4 / 8 -> 486 / 510 / 528 / 450 GSamples/s (at 8 the
				 * count kernel's 128 work-groups no longer keep up with the FFT kernel) */
				for (int want = self->kn_frame_group; want >= 1; want >>= 1)
					if (want <= 32 && !(want & (want - 1)) && sub_c % want == 0 && cpb % want == 0) {
						G = want;
						break;
					}
			}
			const int use_alt = self->overlap && self->alt && !(self->log2n == 16 && self->k1h_fused);
			int used_alt = 0;
			if (drain_h_sets(self, st2))
				return -EIO;
			if (use_alt && (did_prep || !self->relaxed)) {
				if (k1_streams_fork(self))
					return -EIO;
			}
			for (int c0 = 0; c0 < cpb; c0 += sub_c) {
				const int nc = (cpb - c0 < sub_c) ? cpb - c0 : sub_c;
				const int t0 = c0 * 1024, sub_total = nc * 1024;
				hipStream_t ks = self->stream;
				K2Params k2;
				int stores_rows;

				if (use_alt) {
					ks = self->k1_streams[self->k1_seq++ % self->n_k1_streams];
					used_alt = 1;
				}
				set = self->pp;
				self->pp = (self->pp + 1) % self->n_sets;
				self->d_bins = self->d_bins_pp[set];
				self->d_partial = self->d_partial_pp[set];
				if (self->overlap && self->set_used[set])
					HIP_TRY(hipStreamWaitEvent(ks, self->ev_set_free[set], 0), "wait for intermediate set");
				tile = pick_tile(self, sub_total, 1024);
				/* global spectrum index tau = t_offset + t0 + t stores its row iff tau >= total_batch - wf_rows */
				wf_first = total_batch - self->wf_rows - t_offset - t0;
				if (wf_first < 0) wf_first = 0;
				stores_rows = wf_first < sub_total;
				if (!stores_rows) wf_first = sub_total;
				fill_k1(self, &k1, (const char *)d_samples + (size_t)t0 * (hop ? hop : self->n) * sample_bytes, sub_total, tile,
				        (self->wf_pos + t_offset + t0) & (self->wf_rows - 1), wf_first, hop);
				if (stores_rows && wf_enter(self, ks))
					return -EIO;
				prof_begin(self, 0, ks);
				HIP_TRY(launch_k1(k1, ks), "launch fft_bin");
				prof_end(self, ks);
				if (stores_rows && wf_leave(self, ks))
					return -EIO;
				if (self->overlap) {
					HIP_TRY(hipEventRecord(self->ev_k1_done[set], ks), "record K1 done");
					HIP_TRY(hipStreamWaitEvent(st2, self->ev_k1_done[set], 0), "K2 waits for K1");
				}
				memset(&k2, 0, sizeof(k2));
				k2.bins = self->d_bins; k2.partial = self->d_partial;
				k2.hc = self->d_hc + (size_t)self->slot * cells;
				k2.hc16 = self->d_slab16 + (size_t)(c0 / G) * cells;
				k2.n = self->n; k2.bins16 = self->bins16 && self->log2n != 13; k2.bins8p1 = (self->log2n == 13);
				k2.bins9 = (self->log2n == 16); k2.total = sub_total;
				k2.batch = sub_total; k2.chunk = 1024 * G; k2.tile = tile; k2.n_bins = self->n_bins;
				k2.w = 1.0f - self->alpha;
				k2.log2_w = (float)log2((double)(1.0f - self->alpha));
				k2.t_offset = t_offset + t0; k2.weight_batch = total_batch;
				k2.chunk_sum = self->d_chunk_sum + (size_t)(c0 / G) * self->n;
				k2.chunk_max = self->d_chunk_max + (size_t)(c0 / G) * self->n;
				prof_begin(self, 1, st2);
				HIP_TRY(launch_k2(k2, nc / G, st2), "launch count");
				prof_end(self, st2);
				if (self->overlap) {
					HIP_TRY(hipEventRecord(self->ev_set_free[set], st2), "record set free");
					self->set_used[set] = 1;
				}
			}
			{
				K2bParams k2b;
				memset(&k2b, 0, sizeof(k2b));
				k2b.chunk_sum = self->d_chunk_sum; k2b.chunk_max = self->d_chunk_max;
				k2b.live_sum = self->d_live_sum + (size_t)self->slot * self->n;
				k2b.vmax = self->d_vmax + (size_t)self->slot * self->n;
				k2b.n_batches = 1; k2b.cpb = cpb / G; k2b.n = self->n;
				k2b.hc16 = self->d_slab16;
				k2b.hc = self->d_hc + (size_t)self->slot * cells;
				k2b.n_bins = self->n_bins;
				HIP_TRY(launch_k2c(k2b, st2), "launch chunk sum");
			}
			if (used_alt && !self->relaxed) {
				if (k1_streams_join(self))
					return -EIO;
			}
			self->wf_pos = (self->wf_pos + total_batch) & (self->wf_rows - 1);
			self->state = ST_PENDING;
			return 0;
		}
	}

	/* one K1 launch: same two-stream pipeline as run(): K1 on `stream`, K2 (and later the exchange and
	 * fosphor_amd_merge's K3) on `stream2`, intermediates rotating between the sets */
	set = self->pp;
	self->pp = (self->pp + 1) % self->n_sets;
	self->d_bins = self->d_bins_pp[set];
	self->d_partial = self->d_partial_pp[set];
	if (self->overlap && self->set_used[set])
		HIP_TRY(hipStreamWaitEvent(self->stream, self->ev_set_free[set], 0), "wait for intermediate set");

	tile = pick_tile(self, n_local, n_local);
	/* global spectrum index tau = t_offset + t stores its row iff tau >= total_batch - wf_rows */
	wf_first = total_batch - self->wf_rows - t_offset;
	if (wf_first < 0) wf_first = 0;
	fill_k1(self, &k1, d_samples, n_local, tile, (self->wf_pos + t_offset) & (self->wf_rows - 1), wf_first, hop);
	if (wf_enter(self, self->stream))
		return -EIO;
	prof_begin(self, 0, self->stream);
	HIP_TRY(launch_k1(k1, self->stream), "launch fft_bin");
	prof_end(self, self->stream);
	if (wf_leave(self, self->stream))
		return -EIO;

	if (self->overlap) {
		HIP_TRY(hipEventRecord(self->ev_k1_done[set], self->stream), "record K1 done");
		HIP_TRY(hipStreamWaitEvent(st2, self->ev_k1_done[set], 0), "K2 waits for K1");
	}
	if (drain_h_sets(self, st2))
		return -EIO;
	if (run_count(self, 1, n_local, tile, self->slot, t_offset, total_batch, st2))
		return -EIO;
	if (self->overlap) {
		HIP_TRY(hipEventRecord(self->ev_set_free[set], st2), "record set free");
		self->set_used[set] = 1;
	}

	/* the ring advances with the data (host state), so the next frame can be accumulated
	 * before this one is merged */
	self->wf_pos = (self->wf_pos + total_batch) & (self->wf_rows - 1);
	self->state = ST_PENDING;
	return 0;
error:
	return -EIO;
}

extern "C" int fosphor_amd_accumulate_device(struct fosphor *self, const void *d_samples,
                                             int n_local, int t_offset, int total_batch)
{
	return accumulate(self, d_samples, n_local, t_offset, total_batch, 0);
}

/* the same with overlap_cc fused into the read (see fosphor_amd_process_device_overlap): d_samples is this rank's part
 * of the UNEXPANDED stream, (n_local - 1) * N / overlap + N samples starting at the first sample of its first spectrum */
extern "C" int fosphor_amd_accumulate_device_overlap(struct fosphor *self, const void *d_samples,
                                                     int n_local, int t_offset, int total_batch, int overlap)
{
	if (!self || overlap < 1 || overlap > self->n || (self->n % overlap))
		return -EINVAL;
	return accumulate(self, d_samples, n_local, t_offset, total_batch, self->n / overlap);
}

extern "C" int fosphor_amd_set_partial_slot(struct fosphor *self, int slot)
{
	if (!self || slot < 0 || slot >= self->max_batches)
		return -EINVAL;
	self->slot = slot;
	return 0;
}

extern "C" int fosphor_amd_get_partials(struct fosphor *self, struct fosphor_amd_partials *out)
{
	if (!self || !out)
		return -EINVAL;
	out->d_hc = self->d_hc + (size_t)self->slot * self->n_bins * self->n;
	out->d_live_sum = self->d_live_sum + (size_t)self->slot * self->n;
	out->d_max = self->d_vmax + (size_t)self->slot * self->n;
	out->n_hc = self->n_bins * self->n;
	out->n_cols = self->n;
	return 0;
}

extern "C" int fosphor_amd_merge(struct fosphor *self, int total_batch)
{
	if (!self || total_batch < 16)
		return -EINVAL;
	if (prepare(self))
		return -EIO;
	if (run_merge(self, 1, total_batch, self->slot, self->overlap ? self->stream2 : self->stream))
		return -EIO;
	self->last_batches = 1;
	self->last_slot0 = self->slot;
	self->last_hc16 = 0;
	self->state = ST_PENDING;
	return 0;
}

/* ---- native exchange (RCCL over xGMI), fosphor_exchange.cpp --------------- */

/* 1 when the RCCL library can be bound in this process (no communicator, no collective: purely local) */
extern "C" int fosphor_amd_comm_available(void)
{
	return xchg_available();
}

/* ncclCommCount of a communicator made by fosphor_amd_comm_init */
extern "C" int fosphor_amd_comm_count(void *comm)
{
	return comm ? xchg_comm_count(comm) : -EINVAL;
}

/* events around an exchange on its stream, while profiling is on (fosphor_amd_exchange_time) */
static void xprof_begin(struct fosphor *self, hipStream_t st)
{
	if (!self->prof)
		return;
	while (self->xev_used + 2 > self->xev_pool.size()) {
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, dep_event_flags() & ~hipEventDisableTiming) != hipSuccess)
			return;
		self->xev_pool.push_back(e);
	}
	(void)hipEventRecord(self->xev_pool[self->xev_used], st);
}

static void xprof_end(struct fosphor *self, hipStream_t st)
{
	if (!self->prof || self->xev_used + 2 > self->xev_pool.size())
		return;
	if (hipEventRecord(self->xev_pool[self->xev_used + 1], st) == hipSuccess)
		self->xev_used += 2;
}

/* Sum of the durations of the exchanges recorded since the last call (hipEvents on the count/merge stream around the
 * ncclGroup), and how many there were; resets.  0 exchanges when profiling was off. */
extern "C" int fosphor_amd_exchange_time(struct fosphor *self, float *ms_total, int *count)
{
	if (!self || !ms_total || !count)
		return -EINVAL;
	if (sync_all(self))
		return -EIO;
	*ms_total = 0.0f; *count = 0;
	for (size_t i = 0; i + 1 < self->xev_used; i += 2) {
		float t = 0.0f;
		if (hipEventElapsedTime(&t, self->xev_pool[i], self->xev_pool[i + 1]) == hipSuccess) {
			*ms_total += t;
			(*count)++;
		}
	}
	self->xev_used = 0;
	return 0;
}

extern "C" int fosphor_amd_comm_unique_id(void *id128)
{
	return id128 ? xchg_unique_id(id128) : -EINVAL;
}

extern "C" int fosphor_amd_comm_init(void **comm, int world, int rank, const void *id128)
{
	if (!comm || !id128 || world < 1 || rank < 0 || rank >= world)
		return -EINVAL;
	return xchg_comm_init(comm, world, rank, id128);
}

extern "C" int fosphor_amd_comm_destroy(void *comm)
{
	return comm ? xchg_comm_destroy(comm) : -EINVAL;
}

/* Between fosphor_amd_accumulate_device and fosphor_amd_merge: one ncclGroup of three all-reduces over the
 * partial arrays of the current slot, queued on the count/merge stream (behind K2, in front of K3). */
extern "C" int fosphor_amd_exchange(struct fosphor *self, void *comm)
{
	if (!self || !comm)
		return -EINVAL;
	const size_t cells = (size_t)self->n_bins * self->n;
	hipStream_t st = self->overlap ? self->stream2 : self->stream;
	xprof_begin(self, st);
	const int rv = xchg_allreduce3(comm, st,
	                               self->d_hc + (size_t)self->slot * cells, cells,
	                               self->d_live_sum + (size_t)self->slot * self->n, self->d_vmax + (size_t)self->slot * self->n,
	                               (size_t)self->n);
	xprof_end(self, st);
	return rv;
}

/* Frequency-sliced form for large states (128 MiB of counts at 65536 x 512): the counts are
 * reduce-scattered -- rank r owns cells [r C / world, (r + 1) C / world) of the [bin][x] array -- and
 * fosphor_amd_merge_sliced updates only that slice of the histogram; fosphor_amd_gather_state all-gathers the
 * slices when a complete histogram is wanted on every rank (once per draw, not once per exchange). */
extern "C" int fosphor_amd_exchange_sliced(struct fosphor *self, void *comm, int world, int rank)
{
	if (!self || !comm || world < 1 || rank < 0 || rank >= world)
		return -EINVAL;
	const size_t cells = (size_t)self->n_bins * self->n;
	if (cells % (size_t)world)
		return -EINVAL;
	hipStream_t st = self->overlap ? self->stream2 : self->stream;
	xprof_begin(self, st);
	const int rv = xchg_reduce_scatter(comm, st,
	                                   self->d_hc + (size_t)self->slot * cells, cells, world, rank,
	                                   self->d_live_sum + (size_t)self->slot * self->n, self->d_vmax + (size_t)self->slot * self->n,
	                                   (size_t)self->n);
	xprof_end(self, st);
	return rv;
}

extern "C" int fosphor_amd_merge_sliced(struct fosphor *self, int total_batch, int world, int rank)
{
	if (!self || total_batch < 16 || world < 1 || rank < 0 || rank >= world)
		return -EINVAL;
	const size_t cells = (size_t)self->n_bins * self->n;
	if (cells % (size_t)world)
		return -EINVAL;
	if (prepare(self))
		return -EIO;
	const int per = (int)(cells / (size_t)world);
	if (run_merge(self, 1, total_batch, self->slot, self->overlap ? self->stream2 : self->stream, 0, 0,
	              per * rank, per * (rank + 1)))
		return -EIO;
	self->last_batches = 1;
	self->last_slot0 = self->slot;
	self->last_hc16 = 0;
	self->state = ST_PENDING;
	return 0;
}

extern "C" int fosphor_amd_gather_state(struct fosphor *self, void *comm, int world, int rank)
{
	if (!self || !comm || world < 1 || rank < 0 || rank >= world)
		return -EINVAL;
	const size_t cells = (size_t)self->n_bins * self->n;
	if (cells % (size_t)world)
		return -EINVAL;
	hipStream_t st = self->overlap ? self->stream2 : self->stream;
	if (k3_stream_enter(self, st))
		return -EIO;
	self->hot_valid = 0;		/* other ranks' cells arrive: the hot-row flags of the sparse merge no longer describe d_hist */
	return xchg_allgather_f32(comm, st, self->d_hist, cells, world, rank);
}

/* ------------------------------------------------------------------------ */
/* Measurement                                                              */
/* ------------------------------------------------------------------------ */

extern "C" void fosphor_amd_profile(struct fosphor *self, int enable)
{
	self->prof = enable == 2 ? 2 : (enable ? 1 : 0);
}

extern "C" int fosphor_amd_kernel_times(struct fosphor *self, float ms[3], int launches[3])
{
	if (!self)
		return -EINVAL;
	if (sync_all(self))
		return -EIO;
	for (int i = 0; i < 3; i++) { ms[i] = 0.0f; launches[i] = 0; }
	for (size_t i = 0; i + 1 < self->ev_used; i += 2) {
		float t = 0.0f;
		int kind = self->ev_kind[i / 2];
		if (hipEventElapsedTime(&t, self->ev_pool[i], self->ev_pool[i + 1]) == hipSuccess) {
			ms[kind] += t;
			launches[kind]++;
		}
	}
	self->ev_used = 0;
	self->ev_kind.clear();
	return 0;
}

/* Time during which at least one kernel of each kind was running (union of the recorded intervals), from the
 * same events fosphor_amd_kernel_times sums.  K1s of consecutive sub-launches overlap on two streams, so the sum
 * of their individual durations counts the shared time twice; the union is what a launch costs.  Call before
 * fosphor_amd_kernel_times (which resets). */
extern "C" int fosphor_amd_kernel_busy(struct fosphor *self, float busy_ms[3])
{
	if (!self)
		return -EINVAL;
	if (sync_all(self))
		return -EIO;
	for (int kind = 0; kind < 3; kind++) {
		std::vector<std::pair<float, float> > iv;
		for (size_t i = 0; i + 1 < self->ev_used; i += 2) {
			float a = 0.0f, b = 0.0f;
			if (self->ev_kind[i / 2] != kind)
				continue;
			if (hipEventElapsedTime(&a, self->ev_pool[0], self->ev_pool[i]) != hipSuccess ||
			    hipEventElapsedTime(&b, self->ev_pool[0], self->ev_pool[i + 1]) != hipSuccess)
				continue;
			iv.push_back(std::make_pair(a, b));
		}
		std::sort(iv.begin(), iv.end());
		float busy = 0.0f, cur_a = 0.0f, cur_b = -1.0f;
		for (size_t i = 0; i < iv.size(); i++) {
			if (cur_b < cur_a || iv[i].first > cur_b) {
				if (cur_b >= cur_a) busy += cur_b - cur_a;
				cur_a = iv[i].first; cur_b = iv[i].second;
			} else if (iv[i].second > cur_b) {
				cur_b = iv[i].second;
			}
		}
		if (cur_b >= cur_a) busy += cur_b - cur_a;
		busy_ms[kind] = busy;
	}
	return 0;
}

/* Measurement hook: the memory traffic of one K1 launch (same loads, same stores, same order)
 * without its arithmetic, on this instance's buffers; average of `reps` launches in ms.  The
 * intermediates of the current set are overwritten (call between launches, results unaffected:
 * every launch rewrites them before reading). */
extern "C" int fosphor_amd_traffic_twin(struct fosphor *self, const void *d_samples, int n_batches, int batch,
                                        int reps, float *ms_out)
{
	K1Params k1;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	const int total = n_batches * batch;
	float ms = 0.0f;
	if (!self || !d_samples || !ms_out || reps < 1 || total < 16 || total > self->max_spectra || self->log2n != 10 || self->bins16)
		return -EINVAL;
	if (fosphor_amd_finish(self) < 0 || prepare(self))
		return -EIO;
	fill_k1(self, &k1, d_samples, total, pick_tile(self, total, batch), 0, total);
	if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
		return -EIO;
	{
		hipError_t le = hipSuccess;
		for (int i = 0; i < 3 && le == hipSuccess; i++)
			le = launch_k1_traffic_twin(k1, self->stream);
		if (le == hipSuccess) le = hipEventRecord(e0, self->stream);
		for (int i = 0; i < reps && le == hipSuccess; i++)
			le = launch_k1_traffic_twin(k1, self->stream);
		if (le == hipSuccess) le = hipEventRecord(e1, self->stream);
		if (le != hipSuccess) {
			(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
			return -EIO;
		}
	}
	if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
		(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
		return -EIO;
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	*ms_out = ms / (float)reps;
	return 0;
}

/* Placement tuning.  On MI355X the same FFT launch runs in one of two states -- its bare memory traffic takes 98 or 109 us per 512 MiB of
 * IQ -- and which one is a property of the ALLOCATIONS involved: the caller's IQ buffer against this instance's intermediate sets (bin
 * indices + tile partials).  Re-allocating either re-rolls it (about even odds; offsets inside an allocation do not matter: measured up to
 * 1 GiB; tools/twin_state.py).  This call measures the memory twin of the FFT kernel against `d_samples` for every intermediate set and
 * re-allocates a set (keeping the rejected allocations until the end, so that the allocator hands out different memory) until its traffic
 * runs at >= 6.0 TB/s or `max_tries` allocations have been tried, keeping the fastest.  us_before / us_after: the slowest set before and
 * after (NULL allowed).  Returns the number of re-allocations made, or -EINVAL / -EIO.  Results never depend on it. */
extern "C" int fosphor_amd_tune_placement(struct fosphor *self, const void *d_samples, int n_batches, int batch, int max_tries,
                                          float *us_before, float *us_after)
{
	const int total = n_batches * batch;
	if (!self || !d_samples || total < 16 || total > self->max_spectra || self->log2n != 10 || self->bins16 || max_tries < 1)
		return -EINVAL;
	if (fosphor_amd_finish(self) < 0)
		return -EIO;
	const size_t bins_bytes = (size_t)self->max_spectra * self->n * (self->bins16 ? 2 : 1);
	size_t tiles_max = (size_t)self->max_spectra / 4;
	const size_t part_bytes = sizeof(float2) * tiles_max * self->n;
	/* what the twin moves per launch: the IQ, one byte of bin index per sample, the tile partials */
	const double bytes = (double)total * self->n * 9.0 + (double)(total / pick_tile(self, total, batch)) * self->n * 8.0;
	/* (a launch of less than 128 MiB is timed by its launch overhead, not by the memory behind it: measured, nothing replaced) */
	if (bytes < 128.0 * 1048576.0)
		max_tries = 1;
	const float good_ms = (float)(bytes / 6.0e12 * 1e3);
	std::vector<void *> rejected;
	int reallocs = 0;
	float worst_before = 0.0f, worst_after = 0.0f;
	uint32_t *const cur_bins = self->d_bins;
	int cur_set = 0;
	for (int i = 0; i < kSets; i++)
		if (self->d_bins_pp[i] == cur_bins) cur_set = i;
	for (int i = 0; i < self->n_sets; i++) {
		float best = 0.0f;
		for (int t = 0; t < max_tries; t++) {
			uint32_t *nb = self->d_bins_pp[i];
			float2 *np = self->d_partial_pp[i];
			if (t > 0) {
				if (hipMalloc((void **)&nb, bins_bytes) != hipSuccess)
					break;
				if (hipMalloc((void **)&np, part_bytes) != hipSuccess) {
					(void)hipFree(nb);
					break;
				}
			}
			uint32_t *const ob = self->d_bins_pp[i];
			float2 *const op = self->d_partial_pp[i];
			self->d_bins_pp[i] = nb; self->d_partial_pp[i] = np;
			self->d_bins = nb; self->d_partial = np;
			float ms = 0.0f;
			if (fosphor_amd_traffic_twin(self, d_samples, n_batches, batch, 8, &ms) != 0) {
				self->d_bins_pp[i] = ob; self->d_partial_pp[i] = op;
				if (t > 0) { (void)hipFree(nb); (void)hipFree(np); }
				break;
			}
			if (t == 0) {
				best = ms;
				if (ms > worst_before) worst_before = ms;
			} else if (ms < best) {
				rejected.push_back(ob); rejected.push_back(op);		/* the new pair is the better one */
				best = ms;
				reallocs++;
			} else {
				self->d_bins_pp[i] = ob; self->d_partial_pp[i] = op;	/* keep what we had */
				rejected.push_back(nb); rejected.push_back(np);
			}
#ifdef FOSPHOR_AMD_PROBES
			static const int tune_all = [] { const char *e = getenv("FOSPHOR_AMD_DBG_TUNE_ALL"); return e ? atoi(e) : 0; }();
#else
			constexpr int tune_all = 0;
#endif
			if (best <= good_ms && t + 1 >= tune_all)
				break;
		}
		if (best > worst_after) worst_after = best;
	}
	for (void *q : rejected)
		(void)hipFree(q);
	self->d_bins = self->d_bins_pp[cur_set];
	self->d_partial = self->d_partial_pp[cur_set];
	if (us_before) *us_before = worst_before * 1e3f;
	if (us_after) *us_after = worst_after * 1e3f;
	return reallocs;
}

/* N = 8192: how many FFT launches ran in the space-sharing form (share_cus work-groups, beside the previous launch's count / merge
 * kernels) and how many took every CU, since the instance was made; cus = the shared form's work-group count (0: this device never
 * shares).  The form is chosen from the state of the queue at submit time, so two runs of the same calls may differ here -- never in
 * their results. */
extern "C" int fosphor_amd_share_stats(struct fosphor *self, long long *shared, long long *full, int *cus)
{
	if (!self)
		return -EINVAL;
	if (shared) *shared = self->k1w_shared;
	if (full) *full = self->k1w_full;
	if (cus) *cus = self->share_cus;
	return 0;
}

extern "C" void *fosphor_amd_stream(struct fosphor *self)
{
	return self ? (void *)self->stream : NULL;
}

extern "C" void *fosphor_amd_upload_stream(struct fosphor *self)
{
	return self ? (void *)(self->copy_stream ? self->copy_stream : self->stream) : NULL;
}

/* private accessor for fosphor_render.cpp (keeps struct fosphor opaque there) */
extern "C" void fosphor_amd_priv_ranges(struct fosphor *self, int *db_ref, int *db_per_div,
                                        double *center, double *span)
{
	*db_ref = self->power.db_ref;
	*db_per_div = self->power.db_per_div;
	*center = self->frequency.center;
	*span = self->frequency.span;
}

/* private accessors for fosphor_cmap.hip */
extern "C" int fosphor_amd_priv_palette(struct fosphor *self, int n, uint32_t **d_palette)
{
	if (!self->d_palette && hipMalloc((void **)&self->d_palette, sizeof(uint32_t) * n) != hipSuccess)
		return -EIO;
	*d_palette = self->d_palette;
	return 0;
}

extern "C" void fosphor_amd_priv_power(struct fosphor *self, float *scale, float *offset)
{
	*scale = self->power.scale;
	*offset = self->power.offset;
}

/* ------------------------------------------------------------------------ */
/* Host-side tables (no GPU)                                                */
/* ------------------------------------------------------------------------ */

extern "C" int fosphor_amd_host_thresholds(int n_bins, float histo_scale, float histo_offset, double *out)
{
	if (!out || n_bins < 2 || n_bins > 65536)
		return -EINVAL;
	build_thresholds(out, n_bins, histo_scale, histo_offset);
	return 0;
}


extern "C" int fosphor_amd_host_twiddle_count(void) { return build_twiddles(NULL, 10, NULL); }

extern "C" int fosphor_amd_host_twiddles(float *out)
{
	if (!out)
		return -EINVAL;
	build_twiddles((float2 *)out, 10, NULL);
	return 0;
}

/* debug: copy the K1 phase-timing accumulators (K1_TIMING builds) to the host */
extern "C" int fosphor_amd_debug_k1_timing(struct fosphor *self, long long *out, int n)
{
	if (!self || !self->d_dbg || !out || n > 16 * 4 * kK1MaxBlocks)
		return -EINVAL;
	if (fosphor_amd_finish(self) < 0)
		return -EIO;
	return hipMemcpy(out, self->d_dbg, sizeof(long long) * n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -EIO;
}

/* Two-stream pipelining of the process paths on (1, default) or off (0: K1, K2, K3 in order on
 * one stream).  Drains queued work first.  Results are identical either way. */
extern "C" int fosphor_amd_set_overlap(struct fosphor *self, int enable)
{
	if (!self)
		return -EINVAL;
	if (fosphor_amd_finish(self) < 0)
		return -EIO;
	self->overlap = enable ? 1 : 0;
	for (int i = 0; i < kSets; i++)
		self->set_used[i] = 0;
	self->hset_used[0] = self->hset_used[1] = 0;
	return 0;
}

/* strict = 1 (default): a device-resident call behaves like work queued on `stream`: its K1s start after what
 * the caller queued there before the call, and what the caller queues there afterwards starts after them.
 * strict = 0: no ordering against `stream` in either direction (the caller guarantees the samples are complete
 * before the call and keeps them until fosphor_amd_finish() or fosphor_amd_wait_input()); consecutive calls then
 * overlap at their edges like the sub-launches inside one call. */
extern "C" int fosphor_amd_set_input_ordering(struct fosphor *self, int strict)
{
	if (!self)
		return -EINVAL;
	self->relaxed = strict ? 0 : 1;
	return 0;
}

/* Makes `stream` wait for every K1 queued so far (the readers of the callers' sample buffers). */
extern "C" int fosphor_amd_wait_input(struct fosphor *self)
{
	if (!self)
		return -EINVAL;
	return k1_streams_join(self);
}

/* The stream K2 / K3 run on: a second stream when the two-stream pipeline is on, else the main
 * one.  A multi-GPU caller enqueues its all-reduce of the partial arrays relative to THIS stream
 * (after fosphor_amd_accumulate_device, before fosphor_amd_merge). */
extern "C" void *fosphor_amd_stream2(struct fosphor *self)
{
	if (!self)
		return NULL;
	return (void *)(self->overlap ? self->stream2 : self->stream);
}
