/*
 * fosphor_amd.h -- MI355X extensions next to the drop-in API of fosphor.h
 *
 * C ABI only: plain pointers and sizes, no C++/torch types.  These entry points
 * are what the severed CL<->GL interop requires (SURVEY 8b "new export"):
 * the reference kept its results in GL objects or in private host mirrors
 * (private.h:40-42, cl.c:1003-1049); here they are plain HBM buffers.
 *
 * Pipeline per launch:
 *   K1 fft_bin   IQ -> windowed Stockham FFT (fft.cl:397-466, bit-identical
 *                arithmetic) -> exact histogram bin index per sample (u8),
 *                waterfall rows, per-tile live/max partials
 *   K2 count     bin indices -> integer hit counts hc[bin][x] (display.cl:161-177)
 *                + per-batch live sum / max per column (display.cl:139,149-150)
 *   K3 merge     histogram rise/decay, live EMA, max-hold (display.cl:186-310)
 * Between K2 and K3 the three arrays {hc (u32, sum), live_sum (f32, sum),
 * max (f32, max)} are exactly what a multi-GPU run all-reduces.
 */
#ifndef FOSPHOR_AMD_H
#define FOSPHOR_AMD_H

#include <stdint.h>

#include "fosphor.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry and kernel constants.  Zero / negative fields take the reference
 * defaults: fft_len_log 10 (private.h:21), n_bins 128 (display.cl:96),
 * wf_rows 1024 (cl.c:528), t0r 16, t0d 1024, alpha 0.002 (cl.c:714-716). */
#define FOSPHOR_AMD_IQ_FP32 0	/* interleaved float (re, im), 8 B per sample: the reference's format */
#define FOSPHOR_AMD_IQ_FP16 1	/* interleaved IEEE half (re, im), 4 B per sample; fft_len_log = 16 only */

struct fosphor_amd_config
{
	int   fft_len_log;	/* 10 (the reference's), 13, or 16 */
	int   n_bins;		/* 16..512, multiple of 16 */
	int   wf_rows;		/* power of two */
	float t0r, t0d, alpha;
	int   device;		/* HIP device ordinal; -1 = current */
	int   max_spectra;	/* capacity of one launch (all batches together); 0 = 1024 */
	int   max_batches;	/* most batches in one launch; 0 = max(8, max_spectra/1024) */
	void *stream;		/* hipStream_t to run on; NULL = create a private one */
	int   iq_format;	/* FOSPHOR_AMD_IQ_*: format of every sample buffer handed to this instance */
};

/* fosphor_init with explicit geometry.  NULL on failure (message on stderr). */
struct fosphor *fosphor_amd_init(const struct fosphor_amd_config *cfg);

/* Process IQ that is ALREADY in device memory: n_batches consecutive batches of
 * `batch` spectra each (batch % 16 == 0), interleaved fp32 (re, im).
 * Semantics are identical to n_batches successive fosphor_process() calls of
 * batch*N samples each (cl.c:870-968) -- every batch gets its own histogram /
 * live / max-hold update in order -- but the FFT+binning of all batches runs
 * as one launch.  Waterfall rows that a later spectrum of the same call
 * overwrites are not stored (the ring ends in the same state).
 * batch is not capped at 1024: the reference's cap is host-side only
 * (cl.c:885); a larger batch means one display launch with that fft_batch.
 * Returns 0, -EINVAL (bad sizes / over capacity), -EIO (device error). */
int fosphor_amd_process_device(struct fosphor *self, const void *d_samples,
                               int n_batches, int batch);

/* Same, with the overlap of the reference's overlap_cc block (lib/overlap_cc_impl.cc:48-79,
 * include/gnuradio/fosphor/overlap_cc.h:24-32) fused into the read: d_samples is the
 * UNEXPANDED stream; spectrum t of the call is samples [t*N/overlap, t*N/overlap + N), so
 * the buffer must hold (n_batches*batch - 1)*N/overlap + N samples.  overlap must divide N;
 * overlap = 1 is fosphor_amd_process_device.  Results equal processing the stream that
 * overlap_cc(N, overlap) would have produced.
 * (fft_len_log = 13: a call whose spectra count is a multiple of 14 336 = 224 tiles of 64 -- e.g. 14 or 28 batches of 4096 --
 * issued while the previous call is still being counted lets the FFT kernel run on 224 CUs and the count / merge kernels of the
 * previous launch on the other 32: +10 % throughput, identical results; DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8.) */
int fosphor_amd_process_device_overlap(struct fosphor *self, const void *d_samples,
                                       int n_batches, int batch, int overlap);

/* Wait for queued work; counterpart of fosphor_cl_finish (cl.c:970-1061):
 * 1 = new results, 0 = nothing was pending, -EIO = device error. */
int fosphor_amd_finish(struct fosphor *self);

/* Plain-buffer view of the results.  Layouts are the reference's host mirrors
 * (private.h:40-42, fosphor.c:54-56), generalised to the configured geometry:
 *   waterfall  float[wf_rows][N]   x = unshifted FFT bin (DC at 0), ring in y
 *   histogram  float[n_bins][N]    row = dB bin (0 = bottom), x unshifted
 *   spectrum   float[2][N][2]      live then max-hold; (x, y) vertices,
 *                                  index = bin ^ N/2 (fft-shifted)
 *   hitcount   uint32[n_bins][N]   integer counts of the LAST batch processed (the pipeline hands 16-bit
 *                                  counts from kernel to kernel; this view of them is written by this call,
 *                                  which waits for it)
 * Pointers are device pointers.  histogram / spectrum / hitcount stay where they are until fosphor_release;
 * d_waterfall is one of two rings and must be re-queried after every process call (a call that rewrites
 * every row of the ring does so in the other one, see fosphor_amd_set_input_ordering).
 * At fft_len_log = 16 the waterfall rings are UNCACHED device memory (hipDeviceMallocUncached: the kernel's row stores must not
 * pass through the L2 that holds its intermediate, DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8): any kernel or copy may read them, reads simply are not
 * cached. */
struct fosphor_amd_buffers
{
	float    *d_waterfall;
	float    *d_histogram;
	float    *d_spectrum;
	uint32_t *d_hitcount;
	int       waterfall_pos;	/* cl.c:1073-1079 */
	int       fft_len, n_bins, wf_rows;
	float     histo_scale, histo_offset;	/* cl.c:1087-1088 */
};
int fosphor_amd_get_buffers(struct fosphor *self, struct fosphor_amd_buffers *out);
/* The same without the hit-count view: d_hitcount is NULL, no kernel is launched and nothing is waited for (for
 * front ends that poll waterfall_pos / the pointers per frame). */
int fosphor_amd_get_buffers_nohc(struct fosphor *self, struct fosphor_amd_buffers *out);

/* Host copies (synchronise first).  which: 0 waterfall, 1 histogram, 2 spectrum,
 * 3 hitcount.  `bytes` must equal the buffer size.  0 / -EINVAL / -EIO. */
int fosphor_amd_read(struct fosphor *self, int which, void *host, uint64_t bytes);

/* Kernel-level test hook: windowed forward FFT only (the fft1D_1024 contract,
 * fft.cl:397-466): d_in, d_out are float2[n_spectra][N] device buffers; n_spectra a multiple of 4
 * (of 8 at fft_len_log = 13), at most max_spectra; -EINVAL otherwise. */
int fosphor_amd_fft(struct fosphor *self, const void *d_in, void *d_out, int n_spectra);

/* Kernel-level test hook: per-sample bin index and approximate log-power of FFT
 * outputs already in device memory: d_fft float2[n], d_pwr float[n], d_bin uint8[n] -- or
 * uint16[n] on instances with 16-bit bin indices (n_bins > 256 or fft_len_log != 10). */
int fosphor_amd_bin(struct fosphor *self, const void *d_fft, void *d_bin, void *d_pwr, int n);

/* ---- multi-GPU split (one process per GPU; exchange done by the caller) --- */

/* Rank-local half of ONE batch of `total_batch` spectra that is sharded over
 * ranks in contiguous time blocks: this rank holds spectra
 * [t_offset, t_offset + n_local).  Runs K1+K2 and leaves the three partial
 * arrays in device memory; nothing persistent is updated except this rank's
 * waterfall rows and the ring position (advanced by total_batch). */
int fosphor_amd_accumulate_device(struct fosphor *self, const void *d_samples,
                                  int n_local, int t_offset, int total_batch);

/* The same with the overlap of overlap_cc fused into the read (fosphor_amd_process_device_overlap): d_samples is this
 * rank's part of the UNEXPANDED stream, (n_local - 1) * N / overlap + N samples from the first sample of its first
 * spectrum. */
int fosphor_amd_accumulate_device_overlap(struct fosphor *self, const void *d_samples,
                                          int n_local, int t_offset, int total_batch, int overlap);

/* Select which of the instance's partial-array slots (0 .. max_batches - 1) the next
 * accumulate / get_partials / merge use.  Two slots let the all-reduce of frame k overlap the
 * FFT of frame k+1. */
int fosphor_amd_set_partial_slot(struct fosphor *self, int slot);

/* The partial arrays to all-reduce: hc uint32[n_bins][N] (sum),
 * live_sum float[N] (sum), max float[N] (max). */
struct fosphor_amd_partials
{
	uint32_t *d_hc;
	float    *d_live_sum;
	float    *d_max;
	int       n_hc, n_cols;
};
int fosphor_amd_get_partials(struct fosphor *self, struct fosphor_amd_partials *out);

/* Apply K3 with the (reduced) partial arrays as one batch of total_batch spectra. */
int fosphor_amd_merge(struct fosphor *self, int total_batch);

/* ---- the exchange itself, native: RCCL over xGMI (bound at run time, so single-GPU users need no RCCL) ----
 *
 * One communicator per process / GPU.  Rank 0 obtains a 128-byte id and hands it to the other ranks by any
 * means (MPI, a file, torch.distributed); every rank then calls fosphor_amd_comm_init with the same id.
 * Per display frame a rank calls
 *     fosphor_amd_accumulate_device(...)      K1 on `stream`, K2 on the count/merge stream
 *     fosphor_amd_exchange(self, comm)        ncclGroupStart; AllReduce(hc, u32, sum); AllReduce(live sum, f32,
 *                                             sum); AllReduce(max, f32, max); ncclGroupEnd -- same stream
 *     fosphor_amd_merge(self, total_batch)    K3, same stream
 * and none of the three waits on the host: the exchange of frame k overlaps K1 of frame k + 1.
 * Integer sums are exact and order-independent, so the reduced counts are bit-identical on every rank and to a
 * single-GPU launch with fft_batch = total_batch; every rank then applies the identical K3 to identical
 * inputs (replicated state).  0 / -EINVAL / -EIO / -ENOSYS (no RCCL library found). */
int fosphor_amd_comm_unique_id(void *id128);
int fosphor_amd_comm_init(void **comm, int world, int rank, const void *id128);	/* on the current HIP device */
int fosphor_amd_comm_destroy(void *comm);
int fosphor_amd_exchange(struct fosphor *self, void *comm);
/* 1 when an RCCL library can be bound in this process, else 0.  Purely local (no communicator, no collective):
 * ranks agree on it BEFORE any of them enters the id hand-off or ncclCommInitRank. */
int fosphor_amd_comm_available(void);
/* ncclCommCount of a communicator made by fosphor_amd_comm_init: the ranks RCCL itself says it spans (< 0: error) */
int fosphor_amd_comm_count(void *comm);
/* While fosphor_amd_profile() is on, every exchange is bracketed by hipEvents on the stream it runs on: sum of their
 * durations in ms and their number since the last call (resets; waits for the instance's streams). */
int fosphor_amd_exchange_time(struct fosphor *self, float *ms_total, int *count);

/* Frequency-sliced form for large states (65536 x 512: 128 MiB of counts): the counts are reduce-scattered,
 * rank r owning cells [r C / world, (r + 1) C / world) of the [bin][x] arrays (C = n_bins * N must divide), and
 * fosphor_amd_merge_sliced updates only that slice of the histogram (live / max-hold columns: everywhere).
 * Half the bytes of an all-reduce per exchange and 1 / world of the merge per rank.  The histogram is then
 * complete on a rank only inside its slice until fosphor_amd_gather_state all-gathers it (once per draw, if a
 * single-device view is wanted at all); the hitcount view is valid inside the slice only. */
int fosphor_amd_exchange_sliced(struct fosphor *self, void *comm, int world, int rank);
int fosphor_amd_merge_sliced(struct fosphor *self, int total_batch, int world, int rank);
int fosphor_amd_gather_state(struct fosphor *self, void *comm, int world, int rank);

/* ---- measurement ---------------------------------------------------------- */

/* enable = 1: every K1/K2/K3 launch is bracketed by hipEvents on the stream it
 * runs on; enable = 2: K1 launches only (events beside K2/K3 lengthen the
 * count/merge streams' critical path by a few microseconds per launch);
 * 0: off.  fosphor_amd_kernel_times synchronises, returns the summed
 * milliseconds and launch counts per kernel since the last call, and resets. */
void fosphor_amd_profile(struct fosphor *self, int enable);
int  fosphor_amd_kernel_times(struct fosphor *self, float ms[3], int launches[3]);
/* Milliseconds during which at least one K1 / K2 / K3 was running (union of the same intervals; K1s of
 * consecutive sub-launches overlap on two streams).  Call before fosphor_amd_kernel_times, which resets. */
int  fosphor_amd_kernel_busy(struct fosphor *self, float busy_ms[3]);

/* The memory traffic of one K1 launch over (d_samples, n_batches, batch) -- its loads in its order
 * with its prefetch depth, its stores -- without the arithmetic: the practical floor the memory
 * system sets for that launch.  Average of `reps` launches, in milliseconds.  N = 1024, <= 256 bins.
 * State and results of the instance are not changed.  0, -EINVAL, -EIO. */
int fosphor_amd_traffic_twin(struct fosphor *self, const void *d_samples, int n_batches, int batch,
                             int reps, float *ms_out);

/* ---- host-side tables (no GPU needed; exported for tests and for front ends) ---- */

/* The exact bin thresholds the kernels compare |X|^2 against: out[n_bins + 1] doubles.
 * out[b] (1 <= b < n_bins) = smallest s with bin(s) >= b under the pinned pipeline
 * log10(hypot()) -> round(scale * (pwr + offset)) of display.cl:136,161-168;
 * out[0] = -1; out[n_bins] = smallest s whose float hypot overflows. */
int fosphor_amd_host_thresholds(int n_bins, float histo_scale, float histo_offset, double *out);

/* The FFT twiddle table the kernels use (fft.cl:62-68,162-166,286-297 with the pinned
 * sin/cos): out[2 * fosphor_amd_host_twiddle_count()] floats, (cos, sin) pairs laid out
 * pass 2 [k<8][n=1..7], pass 3 [k<64][n=1..7], pass 4 [k<512]. */
int fosphor_amd_host_twiddle_count(void);
int fosphor_amd_host_twiddles(float *out);

/* Pipelining of fosphor_process / fosphor_amd_process_device: with overlap on (default) K2/K3
 * of launch i run on a second stream next to K1 of launch i+1.  0 = strictly one stream.
 * Results are identical; used by the bench to time K1 in isolation. */
int fosphor_amd_set_overlap(struct fosphor *self, int enable);

/* Ordering of fosphor_amd_process_device* calls against `stream` (the stream of the config, or the private one).
 * A call with more samples than one sub-launch (64 Mi by default) is cut into sub-launches whose K1s alternate
 * between `stream` and a second private stream, so that consecutive K1s overlap at their edges.
 * strict = 1 (default): the call still behaves like work queued on `stream` -- its K1s start after what the
 * caller queued there before the call, and what the caller queues there afterwards starts after them.
 * strict = 0: no ordering against `stream` in either direction: the caller guarantees that the samples are
 * complete before the call and keeps them intact until fosphor_amd_finish() or until work it queues behind
 * fosphor_amd_wait_input(); consecutive CALLS then overlap at their edges as well.  Results are identical. */
int fosphor_amd_set_input_ordering(struct fosphor *self, int strict);

/* Makes `stream` wait (device-side, the host does not block) for every K1 queued so far, i.e. for the readers of
 * all sample buffers handed to fosphor_amd_process_device* up to now.  0 / -EINVAL / -EIO. */
int fosphor_amd_wait_input(struct fosphor *self);

/* hipStream_t the instance runs on (K1), and the one K2 / K3 run on (a second stream while the
 * pipeline is on).  The all-reduce between accumulate and merge must be ordered on the latter. */
void *fosphor_amd_stream(struct fosphor *self);
void *fosphor_amd_stream2(struct fosphor *self);
/* hipStream_t the uploads of fosphor_amd_process_pinned are queued on (the instance's own stream until the first such
 * call): an event recorded here behind a call completes when that call's samples have left the caller's buffer --
 * replaces the reference's blocking upload, lib/fosphor/cl.c:903-910. */
void *fosphor_amd_upload_stream(struct fosphor *self);

/* Placement tuning (1024-point instances with 8-bit bin indices).  On MI355X one and the same FFT launch runs in one of two
 * states -- its bare memory traffic takes 98 or 109 us per 512 MiB of IQ -- and which one is decided by the ALLOCATIONS involved:
 * the caller's IQ buffer against the instance's intermediate sets.  Re-allocating either side re-rolls it at about even odds.
 * This call times the FFT kernel's memory traffic against `d_samples` (n_batches x batch spectra, as for
 * fosphor_amd_process_device) for every intermediate set and re-allocates a set until its traffic runs at 6 TB/s or
 * `max_tries` allocations have been tried, keeping the fastest.  *us_before / *us_after: the slowest set's time per launch
 * before and after (NULL allowed).  Returns the number of sets' allocations replaced, -EINVAL or -EIO.  Results never depend on
 * it; call it once, after the instance and the input buffer exist (a few milliseconds). */
int fosphor_amd_tune_placement(struct fosphor *self, const void *d_samples, int n_batches, int batch, int max_tries,
                               float *us_before, float *us_after);

/* Host-logic test hook (no device needed): batches per sub-launch of a fosphor_amd_process_device* call of n_batches batches of
 * `batch` spectra at FFT length 2^fft_len_log, for a sub-launch size of sub_samples samples (64 Mi by default, 1 Gi at
 * fft_len_log = 13) -- at fft_len_log = 13 with the streams on, whole multiples of the unit that lets a piece share the chip
 * (224 tiles of 64 spectra: the split of a 256-CU device; advisory on any other, where no launch shares and the unit only shapes the
 * pieces).  -EINVAL for nonsense. */
int fosphor_amd_plan_piece_batches(int fft_len_log, int overlap, int n_batches, int batch, long long sub_samples);

/* FFT launches made in the space-sharing form (*cus work-groups, the count / merge kernels of the launch before on the CUs they
 * leave) and in the full-chip form since the instance was made: counted at fft_len_log = 13 only (the one length that shares; both
 * counters stay 0 otherwise).  *cus is the device's share whatever the length: 224 on a 256-CU device, 0 on any other (this device
 * never shares: the split is laid out for 256 CUs).  Which form a launch takes depends on what is still queued when it is
 * submitted, never on the data; results are identical (tests: test_c3_space_sharing_*).  Any pointer may be NULL. */
int fosphor_amd_share_stats(struct fosphor *self, long long *shared, long long *full, int *cus);

/* Library identification: "fosphor_amd <version> gfx950". */
const char *fosphor_amd_version(void);

#ifdef __cplusplus
}
#endif

#endif /* FOSPHOR_AMD_H */
