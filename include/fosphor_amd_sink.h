/*
 * fosphor_amd_sink.h -- GNU-Radio-free sink runtime around the HIP compute core
 *
 * Mirrors the data path of gr::fosphor::base_sink_c_impl (lib/base_sink_c_impl.{h,cc}) and
 * gr::fosphor::fifo (lib/fifo.{h,cc}) without GNU Radio, GL or a window system, so that the
 * streaming behaviour (work() -> fifo -> worker thread -> fosphor_process / fosphor_draw) can be
 * built, tested and measured where GNU Radio is absent.  A GNU Radio block wraps it 1:1:
 * base_sink_c::work() forwards to sink_runtime::work(), the GUI shells forward their key /
 * mouse callbacks to execute_ui_action().  INTEGRATION.md shows the wrapper.
 *
 * Differences from the reference, all on purpose:
 *   - the FIFO lives in pinned host memory and the uploads DMA straight out of it, on a stream of
 *     their own; several regions are in flight at once and a region is read_discard()ed only when the
 *     event behind ITS copy has completed (checked without blocking).  A pass of render() queues the
 *     kernels of what the pass before uploaded (fosphor_amd_process_uploaded), then the next uploads
 *     (fosphor_amd_upload_pinned), then draws: the per-frame fosphor_draw waits for kernels only and
 *     the link works on across it.  The reference enqueues a non-blocking write from the FIFO and
 *     discards immediately (cl.c:903-910 vs base_sink_c_impl.cc:168-174): a latent race, not reproduced;
 *   - with a FIFO of 4 Mi samples and more a call carries several whole 1024-spectrum batches (up to 8;
 *     applied one after the other like so many calls): the host's time per call, not the link, bounded
 *     the reference's one-batch-per-call loop (base_sink_c_impl.cc:157-158);
 *   - work() splits copies of 128 Ki samples and more over 8 threads with non-temporal stores (one
 *     core's memcpy is ~1.5 GSamples/s, a PCIe Gen5 x16 link carries 7.9); a source that can write into
 *     the ring itself uses write_prepare() / write_commit() and no host copy is left;
 *   - no GL context: "visible" only decides whether render() synchronises (fosphor_draw) per frame;
 *   - set_fft_window takes the 1024 taps (gr::fft::window::build is GNU Radio's, the caller's).
 */
#ifndef FOSPHOR_AMD_SINK_H
#define FOSPHOR_AMD_SINK_H

#include <stdint.h>

#include "fosphor_amd.h"

#ifdef __cplusplus
#include <atomic>
#include <complex>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace fosphor_amd {

/* lib/fifo.h:20-46 -- the same interface and observable semantics: power-of-two ring of complex samples, one
 * slot kept empty (free = len - 1 - used, fifo.cc:28-38), contiguous zero-copy regions (write_max_size /
 * read_max_size = distance to the end of the ring), blocking prepare / peek.
 * Built differently: a single-producer / single-consumer ring on two monotonic 64-bit counters (samples ever
 * committed, samples ever discarded; position = counter & mask).  used() / free() and the two commit paths are
 * lock-free atomic operations; the mutex and the condition variable exist only to put a blocked side to sleep. */
class fifo
{
 public:
	explicit fifo(int length, bool pinned = false);
	~fifo();
	fifo(const fifo &) = delete;
	fifo &operator=(const fifo &) = delete;

	int free() const { return capacity_ - 1 - used(); }
	int capacity() const { return capacity_; }
	int used() const { return (int)(committed_.load(std::memory_order_seq_cst) - discarded_.load(std::memory_order_seq_cst)); }

	int write_max_size() const { return capacity_ - (int)(committed_.load(std::memory_order_relaxed) & mask_); }
	std::complex<float> *write_prepare(int size, bool wait = true);
	/* write_prepare that gives up: NULL when `size` free slots did not appear within timeout_ms */
	std::complex<float> *write_prepare_for(int size, int timeout_ms);
	void write_commit(int size);

	int read_max_size() const { return capacity_ - (int)(discarded_.load(std::memory_order_relaxed) & mask_); }
	std::complex<float> *read_peek(int size, bool wait = true);
	void read_discard(int size);
	/* regions beyond the read position (the sink keeps several uploads in flight before discarding):
	 * contiguous samples available at `offset` behind the read position, and their address */
	int peek_max_size_at(int offset) const;
	std::complex<float> *peek_at(int offset) const;
	int length() const { return capacity_; }

	bool pinned() const { return pinned_; }

 private:
	template <class Pred> bool sleep_until(Pred ready, int timeout_ms);
	void wake();

	std::complex<float> *ring_;
	int capacity_;
	uint64_t mask_;
	bool pinned_;
	std::atomic<uint64_t> committed_, discarded_;	/* totals since construction */
	std::atomic<int> sleepers_;
	std::mutex sleep_mutex_;
	std::condition_variable sleep_cv_;
};

/* include/gnuradio/fosphor/base_sink_c.h:24-59 + lib/base_sink_c_impl.{h,cc}, data path only */
class sink_runtime
{
 public:
	enum ui_action_t {	/* base_sink_c.h:35-48, same order */
		DB_PER_DIV_UP, DB_PER_DIV_DOWN, REF_UP, REF_DOWN,
		ZOOM_TOGGLE, ZOOM_WIDTH_UP, ZOOM_WIDTH_DOWN, ZOOM_CENTER_UP, ZOOM_CENTER_DOWN,
		RATIO_UP, RATIO_DOWN, FREEZE_TOGGLE,
	};

	explicit sink_runtime(int fifo_length = 2 * 1024 * 1024);	/* base_sink_c_impl.cc:58 */
	~sink_runtime();

	/* base_sink_c_impl::work (base_sink_c_impl.cc:432-462): copies up to noutput_items samples
	 * into the FIFO (blocking while it is full), returns how many were taken. */
	int work(int noutput_items, const std::complex<float> *in);

	bool start();		/* base_sink_c_impl.cc:464-472: spawns the worker */
	bool stop();		/* :474-483; drains what is already in the FIFO first */

	enum mouse_action_t { CLICK };				/* base_sink_c.h:50-52 */

	void execute_ui_action(ui_action_t action);		/* :305-369 */
	/* :371-397: a click inside the main or the zoom pane is turned into the frequency under the cursor; where the
	 * reference publishes it on the block's "freq" message port, this runtime hands it to the callback (if any) and
	 * returns it.  true when a frequency was produced. */
	bool execute_mouse_action(mouse_action_t action, int x, int y, double *freq = nullptr);
	void set_freq_callback(void (*cb)(double freq, void *user), void *user)
	{
		std::lock_guard<std::mutex> lk(d_ui_mutex);		/* (the pair is read as one snapshot by execute_mouse_action) */
		d_freq_cb = cb; d_freq_user = user;
	}
	void reshape(int width, int height);			/* cb_reshape, :291-296 */
	/* copy of a pane layout as the worker last computed it (zoom = false: main pane) */
	struct fosphor_render render_copy(bool zoom) const;
	void set_frequency_range(double center, double span);	/* :399-405 */
	void set_frequency_center(double center);
	void set_frequency_span(double span);
	void set_fft_window(const float *win);			/* :423-430, taps instead of an enum */
	void set_visible(bool visible);				/* cb_visibility, :297-302 */

	struct fosphor *core() { return d_fosphor; }
	uint64_t frames() const { return d_frames.load(); }
	uint64_t samples_processed() const { return d_samples.load(); }
	/* samples taken out of the FIFO but NOT processed because the device refused them (upload / process error other than "busy") */
	uint64_t samples_dropped() const { return d_dropped.load(); }
	int db_ref() const { return ui_snapshot().db_ref; }
	int db_per_div() const { return k_db_per_div[ui_snapshot().db_per_div_idx]; }
	bool frozen() const { return d_frozen.load(); }

 private:
	enum {	/* base_sink_c_impl.h:56-69 */
		SETTING_DIMENSIONS = 1 << 0, SETTING_POWER_RANGE = 1 << 1, SETTING_FREQUENCY_RANGE = 1 << 2,
		SETTING_FFT_WINDOW = 1 << 3, SETTING_RENDER_OPTIONS = 1 << 4,
	};
	static const int k_db_per_div[5];

	void worker();
	void render();
	void retire_uploads(bool wait_all);
	void count_dropped(int len, const char *what, int rv);
	void copy_helper(int idx);
	void settings_mark_changed(uint32_t s) { d_pending.fetch_or(s, std::memory_order_acq_rel); }		/* base_sink_c_impl.cc:204-209 */
	uint32_t settings_get_and_reset_changed() { return d_pending.exchange(0, std::memory_order_acq_rel); }	/* :211-218 */
	void settings_apply(uint32_t s);
	/* everything the UI thread writes and the worker reads, as one value: the worker takes ONE copy per settings pass
	 * under d_ui_mutex and uses only that copy (a ZOOM_TOGGLE between two reads of the live fields would otherwise give
	 * a mixed layout) */
	struct ui_state {
		int width, height, db_ref, db_per_div_idx;
		bool zoom_enabled; double zoom_center, zoom_width; float ratio;
		double freq_center, freq_span;
	};
	ui_state ui_snapshot() const { std::lock_guard<std::mutex> lk(d_ui_mutex); return d_ui; }
	void layout_panes(const ui_state &ui);

	fifo *d_fifo;
	struct fosphor *d_fosphor;
	struct fosphor_render *d_render_main, *d_render_zoom;	/* :61-66 */
	ui_state d_ui;					/* written by the UI thread, under d_ui_mutex */
	mutable std::mutex d_ui_mutex;
	bool d_zoom_applied;				/* worker only: the zoom flag of the layout it last applied */
	void (*d_freq_cb)(double, void *);
	void *d_freq_user;
	std::thread d_worker;
	std::atomic<bool> d_active, d_frozen, d_visible, d_draining;
	std::atomic<uint32_t> d_pending;		/* SETTING_* bits waiting for the worker */
	mutable std::mutex d_render_mutex;		/* the two pane layouts + d_fosphor: worker vs. UI thread (the reference's d_render_mutex) */
	float d_fft_window[1024]; bool d_have_window;
	std::atomic<uint64_t> d_frames, d_samples;

	/* uploads in flight: FIFO regions handed to fosphor_amd_process_pinned and not yet discarded */
	enum { kMaxInflight = 32 };
	struct { void *event; int len; } d_inflight[kMaxInflight];
	int d_inflight_head, d_inflight_n, d_inflight_samples;
	int d_batches_per_call;		/* whole 1024-spectrum batches one fosphor_amd_process_pinned call may carry */
	void *d_events[kMaxInflight];

	/* Helper threads for large copies in work().  The link behind the FIFO carries ~63 GB/s (PCIe Gen5 x16), one core
	 * copies ~12: a 1 Mi-sample work() call is cut into kCopyHelpers + 1 pieces written with non-temporal stores
	 * (the ring is read next by the DMA engine, not by a core).  Hand-off: a generation counter the helpers poll
	 * for a few tens of microseconds after each job -- a streaming producer calls work() back to back, and a
	 * condition-variable wake-up costs as much as a helper's whole share of the copy -- before they go to sleep. */
	enum { kCopyHelpers = 7 };
	struct copy_job { std::complex<float> *dst; const std::complex<float> *src; size_t n; };
	std::thread d_copy_threads[kCopyHelpers];
	copy_job d_copy_jobs[kCopyHelpers];
	std::mutex d_copy_mutex;
	std::condition_variable d_copy_cv;
	std::atomic<int> d_copy_gen, d_copy_pending, d_copy_sleepers;
	std::atomic<bool> d_copy_quit;
	std::atomic<uint64_t> d_dropped;
	bool d_drop_reported;				/* worker only */

 public:
	/* Zero-copy producer interface: a source that can write its samples anywhere (an SDR driver's receive call, a file
	 * read) fills the pinned ring itself and the host copy of work() disappears.  write_prepare returns room for up
	 * to `want` contiguous samples (*got of them; NULL when the sink is not running or nothing frees up within
	 * timeout_ms), write_commit hands `n <= *got` of them to the worker.  Single producer, like work(). */
	std::complex<float> *write_prepare(int want, int *got, int timeout_ms);
	void write_commit(int n);
};

} // namespace fosphor_amd

extern "C" {
#endif /* __cplusplus */

/* ---- C ABI (ctypes / cgo-style bindings and the tests) ---------------------------------- */

/* fosphor_process from PINNED host memory without the staging copy: the H2D is queued straight
 * from `samples`; the caller must keep the region untouched until fosphor_amd_wait_upload()
 * returns.  Same len rules and return codes as fosphor_process (cl.c:882-886). */
int fosphor_amd_process_pinned(struct fosphor *self, const void *samples, int len);
int fosphor_amd_wait_upload(struct fosphor *self);
/* The same in two steps, so that the next upload can be queued before the kernels of this one are waited for (the sink's frame
 * loop): fosphor_amd_upload_pinned queues the H2D alone (len as above, or a whole number of 1024-spectrum batches up to the
 * instance's max_spectra: they are applied one after the other like so many calls; -EBUSY while two uploads are pending),
 * fosphor_amd_process_uploaded queues the kernels of the oldest pending upload (*len_out = its samples; -EINVAL if none).
 * fosphor_amd_process_pinned = both. */
int fosphor_amd_upload_pinned(struct fosphor *self, const void *samples, int len);
int fosphor_amd_process_uploaded(struct fosphor *self, int *len_out);
int fosphor_amd_pending_uploads(struct fosphor *self);

typedef struct fosphor_amd_fifo fosphor_amd_fifo;
fosphor_amd_fifo *fosphor_amd_fifo_new(int length, int pinned);
void  fosphor_amd_fifo_free(fosphor_amd_fifo *f);
int   fosphor_amd_fifo_free_space(fosphor_amd_fifo *f);
int   fosphor_amd_fifo_used(fosphor_amd_fifo *f);
int   fosphor_amd_fifo_write_max_size(fosphor_amd_fifo *f);
void *fosphor_amd_fifo_write_prepare(fosphor_amd_fifo *f, int size, int wait);
void  fosphor_amd_fifo_write_commit(fosphor_amd_fifo *f, int size);
int   fosphor_amd_fifo_read_max_size(fosphor_amd_fifo *f);
void *fosphor_amd_fifo_read_peek(fosphor_amd_fifo *f, int size, int wait);
void  fosphor_amd_fifo_read_discard(fosphor_amd_fifo *f, int size);

typedef struct fosphor_amd_sink fosphor_amd_sink;
fosphor_amd_sink *fosphor_amd_sink_new(void);
/* same with a FIFO of `fifo_length` samples (power of two >= 32 Ki; the reference's is 2 Mi) */
fosphor_amd_sink *fosphor_amd_sink_new_len(int fifo_length);
/* Measurement: feeds `samples` (n complex samples) `repeats` times through work() from the calling thread, in
 * calls of `chunk` samples, waits until every whole 16-spectrum group of it has been processed, and returns the seconds
 * it took (-1.0 if nothing was consumed for 30 s at any point: sink not started, frozen, or device error).  work() is single-producer, like the GR scheduler's calls. */
double fosphor_amd_sink_feed(fosphor_amd_sink *s, const void *samples, int n, int chunk, int repeats);
void  fosphor_amd_sink_free(fosphor_amd_sink *s);
int   fosphor_amd_sink_start(fosphor_amd_sink *s);
int   fosphor_amd_sink_stop(fosphor_amd_sink *s);
/* work(): copies up to n samples into the FIFO and returns how many it took; blocks while the FIFO is full and the
 * worker is consuming, returns 0 when the sink is not running (not started, stopped, or its core failed to initialise). */
int   fosphor_amd_sink_work(fosphor_amd_sink *s, const void *samples, int n);
void  fosphor_amd_sink_ui_action(fosphor_amd_sink *s, int action);
/* cb_reshape (window size in pixels) and execute_mouse_action(CLICK, x, y) of base_sink_c_impl.cc:291-296,371-397:
 * 1 and *freq = the frequency under the cursor when (x, y) lies in the main or the zoom pane, else 0. */
void  fosphor_amd_sink_reshape(fosphor_amd_sink *s, int width, int height);
int   fosphor_amd_sink_mouse_action(fosphor_amd_sink *s, int action, int x, int y, double *freq);
/* where the reference publishes the clicked frequency on the block's "freq" message port (base_sink_c_impl.cc:385,390) this
 * runtime calls `cb(freq, user)` -- on the thread that reported the click, with no lock of the sink held: the callback may
 * call back into the sink (get_render, mouse_action ...).  cb = NULL removes it. */
void  fosphor_amd_sink_set_freq_callback(fosphor_amd_sink *s, void (*cb)(double freq, void *user), void *user);
/* copies of the two pane layouts (main, zoom) as the runtime maintains them */
void  fosphor_amd_sink_get_render(fosphor_amd_sink *s, int zoom, struct fosphor_render *out);
void  fosphor_amd_sink_set_frequency_range(fosphor_amd_sink *s, double center, double span);
void  fosphor_amd_sink_set_fft_window(fosphor_amd_sink *s, const float *win);
void  fosphor_amd_sink_set_visible(fosphor_amd_sink *s, int visible);
/* Zero-copy feed (instead of work()): room for up to `want` contiguous samples in the pinned FIFO (*got of them; NULL and
 * *got = 0 when the sink is not running or no room appears within timeout_ms); the producer writes them in place and
 * commits n <= *got.  The region is DMA'd to the GPU from where it lies: no host copy at all. */
void *fosphor_amd_sink_write_prepare(fosphor_amd_sink *s, int want, int *got, int timeout_ms);
void  fosphor_amd_sink_write_commit(fosphor_amd_sink *s, int n);
struct fosphor *fosphor_amd_sink_core(fosphor_amd_sink *s);
void  fosphor_amd_sink_stats(fosphor_amd_sink *s, uint64_t *frames, uint64_t *samples, int *db_ref, int *db_per_div, int *frozen);
/* samples the worker took from the FIFO and could not process (device error); 0 in a healthy run */
uint64_t fosphor_amd_sink_dropped(fosphor_amd_sink *s);

#ifdef __cplusplus
}
#endif

#endif /* FOSPHOR_AMD_SINK_H */
