/*
 * fosphor_internal.h -- shared between the kernel TU and the C-ABI TU
 */
#ifndef FOSPHOR_INTERNAL_H
#define FOSPHOR_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fosphor_amd {

constexpr int kLog2N = 10;		/* FOSPHOR_FFT_LEN_LOG, private.h:21 */
constexpr int kN     = 1 << kLog2N;

/* Twiddle table layout (float2 entries), generated on the host with the pinned
 * sin/cos of include/fosphor_portable_math.h from the exact float expressions
 * of fft.cl:62-68,162-166,286-297:
 *   [kTw2Off + k*7 + (n-1)]  pass 2 (p=8):  k in [0,8),  n in [1,8)
 *   [kTw3Off + k*7 + (n-1)]  pass 3 (p=64): k in [0,64), n in [1,8)
 *   [kTw4Off + k]            pass 4 (p=512, radix 2): k in [0,512)            */
constexpr int kTw2Off = 0;
constexpr int kTw3Off = kTw2Off + 8 * 7;
constexpr int kTw4Off = kTw3Off + 64 * 7;
constexpr int kTwLen  = kTw4Off + 512;

constexpr int kK1MaxBlocks = 512;	/* 256 CUs x 2 resident work-groups of 4 waves */
constexpr int kK1V2MaxBlocks = 2048;	/* v2: 256 CUs x 8 resident work-groups of 2 waves */

/* K1: IQ -> FFT -> bin index / waterfall row / live+max partials */
struct K1Params {
	const float2 *iq;		/* spectrum t = iq[t*hop .. t*hop + N) */
	int   hop;			/* samples between spectrum starts: N, or N/overlap (overlap_cc_impl.cc:74-76) */
	int   n, log2n;			/* FFT length */
	int   bins16;			/* 16-bit bin indices, 2 spectra per dword (general kernel) */
	int   tw_off[8];		/* twiddle block offsets: radix-8 passes p = 8, 64, ...; then the radix-2 pass
					 * (N = 65536, radix-16 plan: W16 constants, then p = 16, 256, 4096) */
	const float  *win;		/* [N] */
	const float2 *tw;		/* [kTwLen] */
	const double *thr;		/* [n_bins + 1] exact squared-magnitude thresholds */
	uint32_t     *bins;		/* [total/4][N], 4 consecutive spectra packed per dword */
	float2       *partial;		/* [total/tile][N] (live partial, max) */
	float        *wf;		/* [wf_rows][N] */
	float2       *fft_out;		/* test hook, or nullptr */
	long long    *dbg;		/* K1_TIMING builds: [waves][8] cycle accumulators, or nullptr */
	float2       *scratch;		/* variant 4: [64 clusters][N] intermediate spectrum between the two stages */
	int   iq_half;			/* variant 4: the IQ stream is fp16 (re, im) pairs, 4 B per sample */
	uint32_t *sync;			/* variant 4: cluster counters [64][64] */
	uint32_t *sync_err;		/* ... its error word (host-mapped): set when a bounded cluster wait times out */
	int   dbg_k1h;			/* measurement only (FOSPHOR_AMD_DBG_K1H): 1 no cluster waits, 2 no IQ loads, 4 no row / bin stores,
					 * 8 no intermediate stores / loads, 64 no stage-A arithmetic, 128 no stage-B arithmetic / epilogue,
					 * 256 no intermediate loads, 512 no intermediate stores -- results are wrong with any of them */
	int   total;			/* spectra in this launch */
	int   tile;			/* spectra per wave: 4, 8 or 16 */
	int   wf_pos0, wf_mask;		/* ring position of spectrum 0, wf_rows-1 */
	int   wf_first;			/* spectra with index < wf_first do not store their row */
	int   n_bins;
	float binA, binC;		/* v ~= binA * log2(|X|^2) + binC */
	float amb;			/* confident when |v - rint(v)| + kappa |l2| <= amb */
	float kappa;			/* v_log_f32 error bound per unit of |log2 s|, through the slope binA */
	float w;			/* 1 - alpha */
	int   cus;			/* N = 8192: work-groups (= CUs) of the FFT launch when it leaves CUs to the count / merge kernels
					 * (kK1wShareCus; the launch's tiles are a multiple of it), 0 = every CU */
	int   n_cus;			/* CUs of the device (0: 256) */
	int   variant;			/* 1: one wave per spectrum; 2: two waves per spectrum (odd hops); 3: general N (N/8 threads per
					 * spectrum, one LDS slab); 4: N = 65536 in two LDS stages */
};

/* K2: bin indices -> hit counts + live sum / max per column.
 * Grid (N/64 column slabs, chunks): a chunk is `chunk` consecutive spectra of one batch.
 * chunks_per_batch == 1: counts are stored; otherwise they are added with integer atomics
 * into hc[f] (zeroed by the host first) -- integer sums are order-independent, so the result
 * is bit-identical either way.  Float partials go out per chunk and are reduced in a fixed
 * order by k2b_reduce (deterministic). */
struct K2Params {
	const uint32_t *bins;		/* [total/4][N] */
	const float2   *partial;	/* [total/tile][N] */
	uint32_t *hc;			/* [n_batches][n_bins][N] */
	uint16_t *hc16;			/* or (chunk <= 1024): per chunk [n_chunks][N/64][n_bins][32] dwords,
					 * columns c and c + 32 of the slab in the low / high half */
	float    *chunk_sum;		/* [n_chunks][N] */
	float    *chunk_max;		/* [n_chunks][N] */
	int   n;			/* FFT length (columns) */
	int   bins16;			/* bin indices are 16-bit, 2 spectra per dword */
	int   bins8p1;			/* N = 8192: shorts [total / 2][N] (the low bytes of two spectra), then bytes [total / 8][N] (bit u = the 9th bit
					 * of the group's spectrum u): 1.125 B per sample; chunks are multiples of 8 spectra */
	int   bins9;			/* N = 65536: low bytes as [total / 4][N] dwords (4 spectra per dword), then the 9th bits as
					 * [total / tile][N] dwords (bit u = spectrum u of the tile) */
	int   total;			/* spectra of the FFT launch that wrote `bins` (locates the 9th-bit plane) */
	int   batch;			/* spectra per batch in this launch */
	int   chunk;			/* spectra per chunk; divides batch */
	int   tile;
	int   n_bins;
	float w;
	float log2_w;			/* log2(1 - alpha), host double -> float */
	/* sharded batch (multi-GPU): this launch holds spectra [t_offset, t_offset+batch)
	 * of a batch of weight_batch spectra; single GPU: t_offset 0, weight_batch = batch */
	int   t_offset, weight_batch;
	int   dbg_same;			/* measurement only: every chunk reads and writes chunk 0's memory (no HBM traffic) */
	uint32_t *rowmask;		/* hc16 hand-off to K3: [N/64][mask_words][mask_stride >= chunks] one bit per bin row of the slab (the
					 * batches of a row side by side: K3 fetches a row's 64 batches in one request); rows whose 64
					 * counts are all zero are NOT stored and their bit is clear (nullptr: every row is stored) */
	int   mask_words;		/* ceil(n_bins / 32) */
	int   mask_stride;
};

struct K2bParams {
	const float *chunk_sum, *chunk_max;	/* [n_batches * cpb][N] */
	float *live_sum, *vmax;			/* [n_batches][N] */
	int   n_batches, cpb, n;
	/* k2c_sum only (one batch made of cpb chunks whose counts K2 left as packed 16-bit slabs): */
	const uint16_t *hc16;			/* [cpb][N/64][n_bins][32] dwords */
	uint32_t *hc;				/* [n_bins][N] sums */
	int   n_bins;
};

/* K3: histogram rise/decay, live EMA, max-hold */
struct K3Params {
	const uint32_t *hc;		/* [n_batches][n_bins][N] */
	const uint16_t *hc16;		/* or slab-major 16-bit counts, see K2Params */
	uint32_t *hc_export;		/* hc16 path: [n_bins][N] counts of the last batch, for the API view; nullptr: not written
					 * (large states, the view is made on demand from the last batch's slabs) */
	const float    *live_sum;	/* [n_batches][N] */
	const float    *vmax;		/* [n_batches][N] */
	float  *hist;			/* [n_bins][N] */
	float2 *spectrum;		/* [2][N] */
	const float2 *rise;		/* [batch+1] (d, e) per hit count, or nullptr -> computed in-kernel */
	int   n_batches, batch, n_bins, n;
	float t0r, t0d, alpha;
	float live_decay;		/* (1-alpha)^batch */
	int   dbg_same;			/* measurement only: every batch reads batch 0's counts */
	int   cell_begin, cell_end;	/* cells [begin, end) of the (bin, x) array are updated (0, 0 = all): the
					 * frequency-sliced merge of the multi-GPU split; the columns always are */
	const uint32_t *rowmask;	/* hc16 path: K2's row bits (see K2Params) */
	int   mask_words, mask_stride;
	uint8_t *hot;			/* hc16 path: [N/64][n_bins] "some cell of this 64-cell row is above the fast-exit level
					 * (display.cl:237)", maintained here; a row that is not hot and has no count in any batch of
					 * the launch is skipped without reading its cells */
	int   hot_all;			/* the flags are stale (another kernel wrote the histogram): visit every row, rewrite them */
	uint32_t *rowlist;		/* sparse form: [2 + rows] count, the live rows, second count (k3_scan writes, k3_merge reads) */
	int   rowlist_cnt;		/* ... which of the two counts this launch uses: 0, or 1 + rows (alternating) */
};

hipError_t launch_k1(const K1Params &p, hipStream_t s);
hipError_t launch_k1_traffic_twin(const K1Params &p, hipStream_t s);
hipError_t launch_k2(const K2Params &p, int n_chunks, hipStream_t s);
hipError_t launch_k2b(const K2bParams &p, hipStream_t s);
hipError_t launch_k2c(const K2bParams &p, hipStream_t s);
hipError_t launch_k3(const K3Params &p, hipStream_t s);
hipError_t launch_fill(float *dst, float value, size_t n, hipStream_t s);
hipError_t launch_export_hc16(const uint16_t *hc16, const uint32_t *rowmask, int mask_words, int mask_stride, uint32_t *out, int n_bins, int n, hipStream_t s);
hipError_t launch_bin_hook(const float2 *fft, uint8_t *bin, float *pwr, int n,
                           const K1Params &p, int force_exact, hipStream_t s);

/* fosphor_exchange.cpp: RCCL bound at run time */
int xchg_available(void);
int xchg_comm_count(void *comm);
int xchg_unique_id(void *id128);
int xchg_comm_init(void **comm, int world, int rank, const void *id128);
int xchg_comm_destroy(void *comm);
int xchg_allreduce3(void *comm, hipStream_t st, uint32_t *hc, size_t n_hc, float *sum, float *mx, size_t n_cols);
int xchg_reduce_scatter(void *comm, hipStream_t st, uint32_t *hc, size_t n_hc, int world, int rank,
                        float *sum, float *mx, size_t n_cols);
int xchg_allgather_f32(void *comm, hipStream_t st, float *a, size_t n, int world, int rank);

} // namespace fosphor_amd

#endif
