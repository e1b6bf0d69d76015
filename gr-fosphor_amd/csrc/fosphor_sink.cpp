/*
 * fosphor_sink.cpp -- GNU-Radio-free fifo + sink runtime (include/fosphor_amd_sink.h)
 *
 * fifo:          lib/fifo.{h,cc}
 * sink_runtime:  lib/base_sink_c_impl.{h,cc} -- work(), worker(), render(), settings, UI actions
 */
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <chrono>

#include <thread>

#if defined(__x86_64__)
#include <immintrin.h>
#define FOSPHOR_CPU_RELAX() _mm_pause()
#else
#define FOSPHOR_CPU_RELAX() std::this_thread::yield()
#endif

#include <hip/hip_runtime.h>

#include "../../include/fosphor_amd_sink.h"

namespace fosphor_amd {

/* ------------------------------------------------------------------------ */
/* fifo: the surface of lib/fifo.h:20-46 on an SPSC counter ring               */
/* ------------------------------------------------------------------------ */
/* The producer owns committed_, the consumer owns discarded_; each side only ever reads the other's counter.
 * A side that finds too little room / data registers as a sleeper and re-checks under sleep_mutex_ before it
 * waits, and a side that moves its counter takes that mutex only if somebody sleeps -- so the streaming case
 * (neither side blocked) costs two atomic operations per region and no lock. */

fifo::fifo(int length, bool pinned)
  : ring_(nullptr), capacity_(length), mask_((uint64_t)length - 1), pinned_(false),
    committed_(0), discarded_(0), sleepers_(0)
{
	if (pinned) {
		void *p = nullptr;
		if (hipHostMalloc(&p, sizeof(std::complex<float>) * (size_t)length, hipHostMallocDefault) == hipSuccess) {
			ring_ = (std::complex<float> *)p;
			pinned_ = true;
		}
	}
	if (!ring_)
		ring_ = new std::complex<float>[length];
}

fifo::~fifo()
{
	if (pinned_)
		(void)hipHostFree(ring_);
	else
		delete[] ring_;
}

template <class Pred>
bool fifo::sleep_until(Pred ready, int timeout_ms)
{
	if (ready())
		return true;
	std::unique_lock<std::mutex> lk(sleep_mutex_);
	sleepers_.fetch_add(1, std::memory_order_seq_cst);
	bool ok;
	if (timeout_ms < 0) {
		sleep_cv_.wait(lk, ready);
		ok = true;
	} else {
		/* (system_clock: libstdc++ then waits with pthread_cond_timedwait, which ThreadSanitizer intercepts; the steady-clock
		 * form uses pthread_cond_clockwait, which it does not and reports as a double lock.  The time-outs here are
		 * 100 ms slices of a retry loop: a clock step costs one slice.) */
		ok = sleep_cv_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(timeout_ms), ready);
	}
	sleepers_.fetch_sub(1, std::memory_order_seq_cst);
	return ok;
}

void fifo::wake()
{
	if (sleepers_.load(std::memory_order_seq_cst) == 0)
		return;
	{ std::lock_guard<std::mutex> lk(sleep_mutex_); }	/* the sleeper is either before its re-check or inside wait() */
	sleep_cv_.notify_all();
}

std::complex<float> *fifo::write_prepare(int size, bool wait)
{
	if (!sleep_until([&] { return free() >= size; }, wait ? -1 : 0))
		return nullptr;
	return ring_ + (committed_.load(std::memory_order_relaxed) & mask_);
}

std::complex<float> *fifo::write_prepare_for(int size, int timeout_ms)
{
	if (!sleep_until([&] { return free() >= size; }, timeout_ms))
		return nullptr;
	return ring_ + (committed_.load(std::memory_order_relaxed) & mask_);
}

void fifo::write_commit(int size)
{
	committed_.fetch_add((uint64_t)size, std::memory_order_seq_cst);	/* publishes the samples written before it */
	wake();
}

std::complex<float> *fifo::read_peek(int size, bool wait)
{
	if (!sleep_until([&] { return used() >= size; }, wait ? -1 : 0))
		return nullptr;
	return ring_ + (discarded_.load(std::memory_order_relaxed) & mask_);
}

void fifo::read_discard(int size)
{
	discarded_.fetch_add((uint64_t)size, std::memory_order_seq_cst);
	wake();
}

int fifo::peek_max_size_at(int offset) const
{
	const uint64_t at = discarded_.load(std::memory_order_relaxed) + (uint64_t)offset;
	const int64_t avail = (int64_t)(committed_.load(std::memory_order_acquire) - at);
	const int64_t to_end = (int64_t)capacity_ - (int64_t)(at & mask_);
	if (avail <= 0)
		return 0;
	return (int)(avail < to_end ? avail : to_end);
}

std::complex<float> *fifo::peek_at(int offset) const
{
	return ring_ + ((discarded_.load(std::memory_order_relaxed) + (uint64_t)offset) & mask_);
}

/* ------------------------------------------------------------------------ */
/* sink_runtime (lib/base_sink_c_impl.cc)                                   */
/* ------------------------------------------------------------------------ */

const int sink_runtime::k_db_per_div[5] = {1, 2, 5, 10, 20};		/* base_sink_c_impl.cc:48 */

sink_runtime::sink_runtime(int fifo_length)
  : d_fosphor(nullptr), d_zoom_applied(false), d_freq_cb(nullptr), d_freq_user(nullptr),
    d_active(false), d_frozen(false), d_visible(true), d_draining(false),
    d_pending(0),
    d_have_window(false), d_frames(0), d_samples(0),
    d_inflight_head(0), d_inflight_n(0), d_inflight_samples(0), d_batches_per_call(1),
    d_copy_gen(0), d_copy_pending(0), d_copy_sleepers(0), d_copy_quit(false), d_dropped(0), d_drop_reported(false)
{
	d_ui = ui_state{ 1024, 1024, 0, 3, false, 0.5, 0.2, 0.35f, 0.0, 1.0 };
	d_fifo = new fifo(fifo_length, true);				/* base_sink_c_impl.cc:58 */
	d_render_main = new fosphor_render();
	fosphor_render_defaults(d_render_main);				/* :61-62 */
	d_render_zoom = new fosphor_render();
	fosphor_render_defaults(d_render_zoom);				/* :64-66 */
	d_render_zoom->options &= ~(FRO_LABEL_PWR | FRO_LABEL_TIME);
	for (int i = 0; i < kMaxInflight; i++)
		d_events[i] = nullptr;
	for (int i = 0; i < kCopyHelpers; i++)
		d_copy_threads[i] = std::thread(&sink_runtime::copy_helper, this, i);
}

sink_runtime::~sink_runtime()
{
	stop();
	{
		std::lock_guard<std::mutex> lock(d_copy_mutex);
		d_copy_quit.store(true);
	}
	d_copy_cv.notify_all();
	for (int i = 0; i < kCopyHelpers; i++)
		d_copy_threads[i].join();
	for (int i = 0; i < kMaxInflight; i++)
		if (d_events[i])
			(void)hipEventDestroy((hipEvent_t)d_events[i]);
	delete d_render_zoom;
	delete d_render_main;
	delete d_fifo;
}

/* Host copy into the pinned ring with non-temporal stores: the destination is read next by the DMA engine, so the lines
 * need not be fetched for ownership nor kept in the core's caches (a third less memory traffic than memcpy's plain
 * stores, and the source stays cached).  32-byte AVX stores where the CPU has them, memcpy otherwise. */
#if defined(__x86_64__)
__attribute__((target("avx")))
static void stream_copy_avx(void *dst, const void *src, size_t bytes)
{
	char *d = (char *)dst;
	const char *s = (const char *)src;
	const size_t head = (32 - ((uintptr_t)d & 31)) & 31;
	if (bytes < 256 || head > bytes) {
		memcpy(d, s, bytes);
		return;
	}
	memcpy(d, s, head);
	d += head; s += head; bytes -= head;
	size_t i = 0;
	for (; i + 128 <= bytes; i += 128) {
		const __m256i a = _mm256_loadu_si256((const __m256i *)(s + i));
		const __m256i b = _mm256_loadu_si256((const __m256i *)(s + i + 32));
		const __m256i c = _mm256_loadu_si256((const __m256i *)(s + i + 64));
		const __m256i e = _mm256_loadu_si256((const __m256i *)(s + i + 96));
		_mm256_stream_si256((__m256i *)(d + i), a);
		_mm256_stream_si256((__m256i *)(d + i + 32), b);
		_mm256_stream_si256((__m256i *)(d + i + 64), c);
		_mm256_stream_si256((__m256i *)(d + i + 96), e);
	}
	_mm_sfence();
	memcpy(d + i, s + i, bytes - i);
}

#endif

static void stream_copy(void *dst, const void *src, size_t bytes)
{
#if defined(__x86_64__)
	static const bool have_avx = __builtin_cpu_supports("avx");
	if (have_avx) {
		stream_copy_avx(dst, src, bytes);
		return;
	}
#endif
	memcpy(dst, src, bytes);		/* other hosts: plain copy */
}

void sink_runtime::copy_helper(int idx)
{
	int seen = 0;
	for (;;) {
		/* poll for the next generation for ~50 us (a streaming producer is back within that), then sleep */
		const auto t_idle = std::chrono::steady_clock::now();
		int spins = 0;
		while (d_copy_gen.load(std::memory_order_acquire) == seen && !d_copy_quit.load(std::memory_order_relaxed)) {
			FOSPHOR_CPU_RELAX();
			if ((++spins & 255) == 0 &&
			    std::chrono::steady_clock::now() - t_idle > std::chrono::microseconds(50)) {
				std::unique_lock<std::mutex> lock(d_copy_mutex);
				d_copy_sleepers.fetch_add(1, std::memory_order_seq_cst);
				d_copy_cv.wait(lock, [&] { return d_copy_quit.load() || d_copy_gen.load(std::memory_order_acquire) != seen; });
				d_copy_sleepers.fetch_sub(1, std::memory_order_seq_cst);
			}
		}
		if (d_copy_quit.load())
			return;
		seen = d_copy_gen.load(std::memory_order_acquire);
		const copy_job job = d_copy_jobs[idx];		/* written before the generation was published */
		if (job.n)
			stream_copy(job.dst, job.src, sizeof(std::complex<float>) * job.n);
		d_copy_pending.fetch_sub(1, std::memory_order_acq_rel);
	}
}

/* Regions whose upload has completed go back to the producer, oldest first.  wait_all: after the frame's
 * synchronisation point (or at shutdown) everything queued has completed. */
void sink_runtime::retire_uploads(bool wait_all)
{
	while (d_inflight_n) {
		hipEvent_t ev = (hipEvent_t)d_inflight[d_inflight_head].event;
		if (ev) {
			if (wait_all)
				(void)hipEventSynchronize(ev);
			else if (hipEventQuery(ev) != hipSuccess)
				break;
		}
		d_fifo->read_discard(d_inflight[d_inflight_head].len);		/* :174 */
		d_inflight_samples -= d_inflight[d_inflight_head].len;
		d_inflight_head = (d_inflight_head + 1) % kMaxInflight;
		d_inflight_n--;
	}
}

/* The pane layout of base_sink_c_impl.cc:257-289 as data: what each pane gets is decided first (a plain value), then
 * written into the two fosphor_render structs under the render lock and refreshed. */
namespace {
struct pane { int pos_x, width; };
struct split { pane main, zoom; int main_set, main_clear; };

/* zoom on: the main pane takes 65 % of the window and shows the zoom channel instead of the colour scale; the zoom
 * pane starts 10 px inside it and takes the rest.  zoom off: one pane, full width, colour scale back. */
split split_window(int width, bool zoom)
{
	split sp;
	const int cut = zoom ? (int)(width * 0.65f) : width;
	sp.main = pane{ 0, cut };
	sp.zoom = pane{ cut - 10, width - cut + 10 };
	sp.main_set   = zoom ? FRO_CHANNELS : FRO_COLOR_SCALE;
	sp.main_clear = zoom ? FRO_COLOR_SCALE : FRO_CHANNELS;
	return sp;
}
}

void sink_runtime::layout_panes(const ui_state &ui)
{
	const split sp = split_window(ui.width, ui.zoom_enabled);
	std::lock_guard<std::mutex> lk(d_render_mutex);
	struct fosphor_render *both[2] = { d_render_main, d_render_zoom };

	d_render_main->width = sp.main.width;
	d_render_main->options = (d_render_main->options | sp.main_set) & ~sp.main_clear;
	if (ui.zoom_enabled) {				/* the zoom pane keeps its last geometry while it is hidden */
		d_render_zoom->pos_x = sp.zoom.pos_x;
		d_render_zoom->width = sp.zoom.width;
	}
	struct fosphor_channel &ch = d_render_main->channels[0];
	ch.enabled = ui.zoom_enabled;
	ch.center  = (float)ui.zoom_center;
	ch.width   = (float)ui.zoom_width;
	d_render_zoom->freq_center = ch.center;
	d_render_zoom->freq_span   = ch.width;
	for (struct fosphor_render *r : both) {
		r->height = ui.height;
		r->histo_wf_ratio = ui.ratio;
		fosphor_render_refresh(r);
	}
	d_zoom_applied = ui.zoom_enabled;
}

void sink_runtime::settings_apply(uint32_t s)				/* :220-288, compute-relevant part */
{
	if (!s)
		return;
	const ui_state ui = ui_snapshot();				/* one consistent copy for the whole pass */
	if (s & SETTING_POWER_RANGE)
		fosphor_set_power_range(d_fosphor, ui.db_ref, k_db_per_div[ui.db_per_div_idx]);
	if (s & SETTING_FREQUENCY_RANGE)
		fosphor_set_frequency_range(d_fosphor, ui.freq_center, ui.freq_span);
	if ((s & SETTING_FFT_WINDOW) && d_have_window) {
		float win[1024];
		{
			std::lock_guard<std::mutex> lk(d_ui_mutex);
			memcpy(win, d_fft_window, sizeof(win));
		}
		fosphor_set_fft_window(d_fosphor, win);
	}
	if (s & (SETTING_DIMENSIONS | SETTING_RENDER_OPTIONS))
		layout_panes(ui);
}

struct fosphor_render sink_runtime::render_copy(bool zoom) const
{
	std::lock_guard<std::mutex> lk(d_render_mutex);
	return zoom ? *d_render_zoom : *d_render_main;
}

void sink_runtime::reshape(int width, int height)			/* :291-296 */
{
	{
		std::lock_guard<std::mutex> lk(d_ui_mutex);
		d_ui.width = width;
		d_ui.height = height;
	}
	settings_mark_changed(SETTING_DIMENSIONS);
}

bool sink_runtime::execute_mouse_action(mouse_action_t action, int x, int y, double *freq)	/* :371-397 */
{
	if (action != CLICK)
		return false;
	double f;
	{
		/* the worker re-lays the panes and owns d_fosphor's lifetime: the position is mapped under the render lock ... */
		std::lock_guard<std::mutex> lk(d_render_mutex);
		if (!d_fosphor)
			return false;
		const int in_main = fosphor_render_pos_inside(d_render_main, x, y);
		const int in_zoom = d_render_main->channels[0].enabled ? fosphor_render_pos_inside(d_render_zoom, x, y) : 0;
		if (in_main & 1)
			f = fosphor_pos2freq(d_fosphor, d_render_main, x);
		else if (in_zoom & 1)
			f = fosphor_pos2freq(d_fosphor, d_render_zoom, x);
		else
			return false;
	}
	/* ... and published WITHOUT it, like the reference's message_port_pub (:385,390): a callback may re-enter the sink
	 * (get_render, another mouse action) or take its time without stalling the worker */
	if (freq)
		*freq = f;
	void (*cb)(double, void *);
	void *user;
	{ std::lock_guard<std::mutex> lk(d_ui_mutex); cb = d_freq_cb; user = d_freq_user; }	/* the pair as one snapshot */
	if (cb)
		cb(f, user);
	return true;
}

void sink_runtime::worker()						/* :77-122 */
{
	{
		/* the reference's instance (fosphor_init(): one 1024-spectrum batch per call); with a FIFO that can hold several batches,
		 * room for eight per call (render() below) */
		struct fosphor *f;
		int per_call = d_fifo->capacity() / (2 * 1024 * 1024);		/* half the FIFO per call at most: the producer keeps the other half */
		if (per_call > 8) per_call = 8;
		if (per_call >= 2) {
			struct fosphor_amd_config cfg;
			memset(&cfg, 0, sizeof(cfg));
			cfg.max_spectra = per_call * 1024;
			f = fosphor_amd_init(&cfg);
			d_batches_per_call = f ? per_call : 1;
			if (!f)
				f = fosphor_init();
		} else {
			f = fosphor_init();
			d_batches_per_call = 1;
		}
		std::lock_guard<std::mutex> lk(d_render_mutex);
		d_fosphor = f;
	}
	if (!d_fosphor) {
		d_active = false;
		return;
	}
	settings_apply(~(uint32_t)0);					/* :106-109 (+ the pane layout: no window system sends a first reshape here) */
	while (d_active || (d_draining && d_fifo->used() - d_inflight_samples >= 16 * 1024))
		render();
	while (fosphor_amd_pending_uploads(d_fosphor) > 0) {			/* the kernels of the last uploads */
		int plen = 0;
		const int rv = fosphor_amd_process_uploaded(d_fosphor, &plen);
		if (rv == 0)
			d_samples += (uint64_t)plen;
		else
			count_dropped(plen, "kernels of an uploaded region", rv);
	}
	(void)fosphor_amd_finish(d_fosphor);
	retire_uploads(true);
	{
		std::lock_guard<std::mutex> lk(d_render_mutex);	/* no click is being mapped through it */
		fosphor_release(d_fosphor);
		d_fosphor = nullptr;
	}
}

/* A region taken from the FIFO and given back unprocessed (the reference ignores fosphor_process's code, base_sink_c_impl.cc:170): counted,
 * and said once, so that a failing device does not pass for a quiet one -- whichever step failed. */
void sink_runtime::count_dropped(int len, const char *what, int rv)
{
	d_dropped += (uint64_t)(len > 0 ? len : 0);
	if (!d_drop_reported) {
		fprintf(stderr, "[!] fosphor_amd sink: %s failed (%d); samples are being dropped\n", what, rv);
		d_drop_reported = true;
	}
}

void sink_runtime::render()						/* :130-201 */
{
	const int fft_len = 1024, batch_mult = 16, batch_max = 1024, max_iter = 8;
	int i, queued = 0;

	settings_apply(settings_get_and_reset_changed());

	retire_uploads(false);
	/* the kernels of what the previous pass uploaded (pinned FIFO: a pass queues uploads, the next pass their kernels -- the
	 * synchronisation point at the end of a pass then waits for kernels only and the link works on across it) */
	while (fosphor_amd_pending_uploads(d_fosphor) > 0) {
		int plen = 0;
		const int rv = fosphor_amd_process_uploaded(d_fosphor, &plen);
		if (rv == 0)
			d_samples += (uint64_t)plen;
		else
			count_dropped(plen, "kernels of an uploaded region", rv);	/* (uploaded, then not processed: dropped all the same) */
		queued++;
	}
	for (i = 0; i < max_iter; i++) {
		/* the next region starts behind the ones still in flight (they are not discarded yet) */
		int len = d_fifo->peek_max_size_at(d_inflight_samples);
		len &= ~((batch_mult * fft_len) - 1);			/* :156 */
		if (len > (batch_max * fft_len)) {
			/* :157-158 caps a call at one batch.  Here a call may carry up to d_batches_per_call WHOLE batches -- they are applied one
			 * after the other, exactly like that many calls -- because what bounds this loop is the host's time per call (one
			 * upload, a dozen launches and events: ~250 us, against 150 us of DMA per batch) */
			const int nb = len / (batch_max * fft_len);
			len = (nb < d_batches_per_call ? nb : d_batches_per_call) * (batch_max * fft_len);
			if (!d_fifo->pinned())
				len = batch_max * fft_len;
		}
		if (!len)
			break;
		if (d_inflight_n == kMaxInflight)
			retire_uploads(true);
		std::complex<float> *data = d_fifo->peek_at(d_inflight_samples);
		const int slot = (d_inflight_head + d_inflight_n) % kMaxInflight;
		void *ev = nullptr;
		if (!d_frozen) {
			int rv;
			if (d_fifo->pinned()) {
				/* DMA straight from the ring; the region goes back to the producer once the event behind
				 * its copy has completed -- nothing waits here, the next region's copy queues behind it */
				rv = fosphor_amd_upload_pinned(d_fosphor, data, len);
				if (rv == -EBUSY)
					break;				/* both staging buffers hold uploads: their kernels come first (next pass) */
				if (rv)
					count_dropped(len, "upload", rv);
				if (!d_events[slot]) {
					hipEvent_t e;
					if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess)
						d_events[slot] = e;
				}
				if (d_events[slot] && hipEventRecord((hipEvent_t)d_events[slot], (hipStream_t)fosphor_amd_upload_stream(d_fosphor)) == hipSuccess)
					ev = d_events[slot];
				else
					(void)fosphor_amd_wait_upload(d_fosphor);
			} else {
				rv = fosphor_process(d_fosphor, data, len);	/* copies before it returns */
				if (rv == 0)
					d_samples += (uint64_t)len;
				else
					count_dropped(len, "fosphor_process", rv);
			}
		}
		d_inflight[slot].event = ev;
		d_inflight[slot].len = len;
		d_inflight_n++;
		d_inflight_samples += len;
		queued++;
		retire_uploads(false);
	}

	if (d_visible) {
		fosphor_draw(d_fosphor, d_render_main);			/* :178-195: the per-frame sync point */
		if (d_zoom_applied)
			fosphor_draw(d_fosphor, d_render_zoom);		/* :189-190 */
		d_frames++;
		retire_uploads(false);		/* (the uploads queued in this pass run on: their regions go back as their events complete) */
	} else {
		std::this_thread::sleep_for(std::chrono::milliseconds(10));	/* :197-200 */
		retire_uploads(false);
	}
	if (!queued)
		std::this_thread::sleep_for(std::chrono::microseconds(100));	/* nothing queued: do not spin */
}

int sink_runtime::work(int noutput_items, const std::complex<float> *in)	/* :432-462 */
{
	int l = noutput_items;
	int mw = d_fifo->write_max_size();
	if (l > mw)
		l = mw;
	if (!l)
		return 0;
	/* blocks while the FIFO is full -- as long as somebody is consuming: a sink that is not running (never started,
	 * stopped, core failed to initialise) returns 0 instead of blocking for ever */
	std::complex<float> *dst = nullptr;
	while (!dst) {
		if (!d_active)
			return 0;
		dst = d_fifo->write_prepare_for(l, 100);
	}
	if (l >= 128 * 1024) {
		/* one core copies ~12 GB/s; the link behind the FIFO carries 4-5 times that */
		const size_t part = ((size_t)l / (kCopyHelpers + 1)) & ~(size_t)1023;
		for (int k = 0; k < kCopyHelpers; k++) {
			d_copy_jobs[k].dst = dst + part * (k + 1);
			d_copy_jobs[k].src = in + part * (k + 1);
			d_copy_jobs[k].n = (k == kCopyHelpers - 1) ? (size_t)l - part * kCopyHelpers : part;
		}
		d_copy_pending.store(kCopyHelpers, std::memory_order_relaxed);
		d_copy_gen.fetch_add(1, std::memory_order_seq_cst);		/* publishes the jobs */
		if (d_copy_sleepers.load(std::memory_order_seq_cst)) {
			{ std::lock_guard<std::mutex> lock(d_copy_mutex); }	/* a helper between its check and its wait has the mutex */
			d_copy_cv.notify_all();
		}
		stream_copy(dst, in, sizeof(std::complex<float>) * part);
		/* the helpers finish within microseconds of this thread's own part; on an oversubscribed host a helper may have been
		 * descheduled -- after a bounded spin the producer yields its core instead of burning it */
		for (unsigned spins = 0; d_copy_pending.load(std::memory_order_acquire); ) {
			if (++spins < 4096)
				FOSPHOR_CPU_RELAX();
			else
				std::this_thread::yield();
		}
	} else {
		memcpy(dst, in, sizeof(std::complex<float>) * (size_t)l);
	}
	d_fifo->write_commit(l);
	return l;
}

std::complex<float> *sink_runtime::write_prepare(int want, int *got, int timeout_ms)
{
	*got = 0;
	int l = want;
	const int mw = d_fifo->write_max_size();			/* contiguous room up to the end of the ring */
	if (l > mw)
		l = mw;
	if (l <= 0 || !d_active)
		return nullptr;
	std::complex<float> *dst = d_fifo->write_prepare_for(l, timeout_ms);
	if (dst)
		*got = l;
	return dst;
}

void sink_runtime::write_commit(int n)
{
	if (n > 0)
		d_fifo->write_commit(n);
}

bool sink_runtime::start()						/* :464-472 */
{
	if (!d_active) {
		d_active = true;
		d_draining = false;
		d_worker = std::thread(&sink_runtime::worker, this);
	}
	return true;
}

bool sink_runtime::stop()						/* :474-483 */
{
	if (d_active || d_worker.joinable()) {
		d_draining = true;
		d_active = false;
		if (d_worker.joinable())
			d_worker.join();
		d_draining = false;
	}
	return true;
}

void sink_runtime::execute_ui_action(ui_action_t action)		/* :305-369 */
{
	{
		std::lock_guard<std::mutex> lk(d_ui_mutex);
		ui_state &u = d_ui;
		switch (action) {
		case DB_PER_DIV_UP:	if (u.db_per_div_idx < 4) u.db_per_div_idx++; break;
		case DB_PER_DIV_DOWN:	if (u.db_per_div_idx > 0) u.db_per_div_idx--; break;
		case REF_UP:		u.db_ref += k_db_per_div[u.db_per_div_idx]; break;
		case REF_DOWN:		u.db_ref -= k_db_per_div[u.db_per_div_idx]; break;
		case ZOOM_TOGGLE:	u.zoom_enabled = !u.zoom_enabled; break;
		case ZOOM_WIDTH_UP:	if (u.zoom_enabled) u.zoom_width *= 2.0; break;
		case ZOOM_WIDTH_DOWN:	if (u.zoom_enabled) u.zoom_width /= 2.0; break;
		case ZOOM_CENTER_UP:	if (u.zoom_enabled) u.zoom_center += u.zoom_width / 8.0; break;
		case ZOOM_CENTER_DOWN:	if (u.zoom_enabled) u.zoom_center -= u.zoom_width / 8.0; break;
		case RATIO_UP:		if (u.ratio < 0.8f) u.ratio += 0.05f; break;
		case RATIO_DOWN:	if (u.ratio > 0.2f) u.ratio -= 0.05f; break;
		case FREEZE_TOGGLE:	d_frozen = !d_frozen; break;
		}
	}
	settings_mark_changed(SETTING_POWER_RANGE | SETTING_RENDER_OPTIONS);
}

void sink_runtime::set_frequency_range(double center, double span)
{
	{ std::lock_guard<std::mutex> lk(d_ui_mutex); d_ui.freq_center = center; d_ui.freq_span = span; }
	settings_mark_changed(SETTING_FREQUENCY_RANGE);
}
void sink_runtime::set_frequency_center(double center)
{
	{ std::lock_guard<std::mutex> lk(d_ui_mutex); d_ui.freq_center = center; }
	settings_mark_changed(SETTING_FREQUENCY_RANGE);
}
void sink_runtime::set_frequency_span(double span)
{
	{ std::lock_guard<std::mutex> lk(d_ui_mutex); d_ui.freq_span = span; }
	settings_mark_changed(SETTING_FREQUENCY_RANGE);
}
void sink_runtime::set_fft_window(const float *win)
{
	{
		std::lock_guard<std::mutex> lk(d_ui_mutex);
		memcpy(d_fft_window, win, sizeof(d_fft_window));
		d_have_window = true;
	}
	settings_mark_changed(SETTING_FFT_WINDOW);
}
void sink_runtime::set_visible(bool visible) { d_visible = visible; }

} // namespace fosphor_amd

/* ------------------------------------------------------------------------ */
/* C ABI                                                                    */
/* ------------------------------------------------------------------------ */

using fosphor_amd::fifo;
using fosphor_amd::sink_runtime;

struct fosphor_amd_fifo { fifo f; fosphor_amd_fifo(int n, bool p) : f(n, p) {} };
struct fosphor_amd_sink { sink_runtime s; explicit fosphor_amd_sink(int n) : s(n) {} };

extern "C" {

fosphor_amd_fifo *fosphor_amd_fifo_new(int length, int pinned)
{
	if (length < 2 || (length & (length - 1)))
		return nullptr;
	return new fosphor_amd_fifo(length, pinned != 0);
}
void  fosphor_amd_fifo_free(fosphor_amd_fifo *f) { delete f; }
int   fosphor_amd_fifo_free_space(fosphor_amd_fifo *f) { return f->f.free(); }
int   fosphor_amd_fifo_used(fosphor_amd_fifo *f) { return f->f.used(); }
int   fosphor_amd_fifo_write_max_size(fosphor_amd_fifo *f) { return f->f.write_max_size(); }
void *fosphor_amd_fifo_write_prepare(fosphor_amd_fifo *f, int size, int wait) { return f->f.write_prepare(size, wait != 0); }
void  fosphor_amd_fifo_write_commit(fosphor_amd_fifo *f, int size) { f->f.write_commit(size); }
int   fosphor_amd_fifo_read_max_size(fosphor_amd_fifo *f) { return f->f.read_max_size(); }
void *fosphor_amd_fifo_read_peek(fosphor_amd_fifo *f, int size, int wait) { return f->f.read_peek(size, wait != 0); }
void  fosphor_amd_fifo_read_discard(fosphor_amd_fifo *f, int size) { f->f.read_discard(size); }

fosphor_amd_sink *fosphor_amd_sink_new(void) { return new fosphor_amd_sink(2 * 1024 * 1024); }
fosphor_amd_sink *fosphor_amd_sink_new_len(int fifo_length)
{
	if (fifo_length < 32 * 1024 || (fifo_length & (fifo_length - 1)))
		return nullptr;
	return new fosphor_amd_sink(fifo_length);
}
double fosphor_amd_sink_feed(fosphor_amd_sink *s, const void *samples, int n, int chunk, int repeats)
{
	const std::complex<float> *in = (const std::complex<float> *)samples;
	const uint64_t before = s->s.samples_processed();
	/* the sink consumes whole 16-spectrum groups: a remainder stays in the FIFO until more samples arrive */
	const uint64_t want = before + (((uint64_t)n * (uint64_t)repeats) & ~(uint64_t)(16 * 1024 - 1));
	auto t0 = std::chrono::steady_clock::now();
	/* "no progress for 30 s" at any point -- while feeding (the sink is not running, or frozen with a full FIFO) or
	 * while waiting for the tail -- gives up */
	auto t_progress = t0;
	uint64_t seen = before;
	auto stalled = [&]() {
		const uint64_t now_done = s->s.samples_processed();
		const auto now = std::chrono::steady_clock::now();
		if (now_done != seen) { seen = now_done; t_progress = now; }
		return std::chrono::duration<double>(now - t_progress).count() > 30.0;
	};
	for (int r = 0; r < repeats; r++) {
		int pos = 0;
		while (pos < n) {
			int l = n - pos < chunk ? n - pos : chunk;
			int took = s->s.work(l, in + pos);
			pos += took;
			if (took) {
				t_progress = std::chrono::steady_clock::now();
			} else {
				if (stalled())
					return -1.0;
				std::this_thread::sleep_for(std::chrono::microseconds(200));
			}
		}
	}
	while (s->s.samples_processed() < want) {
		std::this_thread::sleep_for(std::chrono::microseconds(50));
		if (stalled())
			return -1.0;			/* the worker is not consuming (stopped, frozen, device error) */
	}
	return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
void  fosphor_amd_sink_free(fosphor_amd_sink *s) { delete s; }
uint64_t fosphor_amd_sink_dropped(fosphor_amd_sink *s) { return s->s.samples_dropped(); }
int   fosphor_amd_sink_start(fosphor_amd_sink *s) { return s->s.start() ? 1 : 0; }
int   fosphor_amd_sink_stop(fosphor_amd_sink *s) { return s->s.stop() ? 1 : 0; }
int   fosphor_amd_sink_work(fosphor_amd_sink *s, const void *samples, int n)
{
	return s->s.work(n, (const std::complex<float> *)samples);
}
void  fosphor_amd_sink_ui_action(fosphor_amd_sink *s, int action) { s->s.execute_ui_action((sink_runtime::ui_action_t)action); }
void  fosphor_amd_sink_reshape(fosphor_amd_sink *s, int w, int h) { s->s.reshape(w, h); }
int   fosphor_amd_sink_mouse_action(fosphor_amd_sink *s, int action, int x, int y, double *freq)
{
	return s->s.execute_mouse_action((sink_runtime::mouse_action_t)action, x, y, freq) ? 1 : 0;
}
void  fosphor_amd_sink_set_freq_callback(fosphor_amd_sink *s, void (*cb)(double, void *), void *user)
{
	s->s.set_freq_callback(cb, user);
}
void  fosphor_amd_sink_get_render(fosphor_amd_sink *s, int zoom, struct fosphor_render *out)
{
	*out = s->s.render_copy(zoom != 0);
}
void  fosphor_amd_sink_set_frequency_range(fosphor_amd_sink *s, double c, double sp) { s->s.set_frequency_range(c, sp); }
void  fosphor_amd_sink_set_fft_window(fosphor_amd_sink *s, const float *win) { s->s.set_fft_window(win); }
void  fosphor_amd_sink_set_visible(fosphor_amd_sink *s, int v) { s->s.set_visible(v != 0); }
void *fosphor_amd_sink_write_prepare(fosphor_amd_sink *s, int want, int *got, int timeout_ms)
{
	int g = 0;
	void *p = s->s.write_prepare(want, &g, timeout_ms);
	if (got) *got = g;
	return p;
}
void  fosphor_amd_sink_write_commit(fosphor_amd_sink *s, int n) { s->s.write_commit(n); }
struct fosphor *fosphor_amd_sink_core(fosphor_amd_sink *s) { return s->s.core(); }
void  fosphor_amd_sink_stats(fosphor_amd_sink *s, uint64_t *frames, uint64_t *samples, int *db_ref, int *db_per_div, int *frozen)
{
	if (frames) *frames = s->s.frames();
	if (samples) *samples = s->s.samples_processed();
	if (db_ref) *db_ref = s->s.db_ref();
	if (db_per_div) *db_per_div = s->s.db_per_div();
	if (frozen) *frozen = s->s.frozen() ? 1 : 0;
}

} // extern "C"
