/*
 * fosphor_oracle.h -- CPU restatement of the fosphor compute hot path
 *
 * TEST INFRASTRUCTURE.  This is the parity oracle for the HIP product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (gr-fosphor_amd/) never links, imports or calls it.
 *
 * It restates, operation for operation, the reference's
 *   lib/fosphor/fft.cl      (windowed Stockham FFT, radix 8.8.8[.8..].2)
 *   lib/fosphor/display.cl  (log-power, waterfall, live EMA, hit counts,
 *                            histogram rise/decay, max-hold with decay)
 *   lib/fosphor/cl.c        (host state machine: first-run fills, waterfall
 *                            ring position, histogram range)
 *   lib/fosphor/fosphor.c   (default window, power range -> scale/offset)
 * with the OpenCL built-ins bound to include/fosphor_portable_math.h
 * (sin, cos, hypot, log10, round) and to IEEE / glibc for the ones that only
 * feed tolerance-checked floats (powf for native_powr, 1.0f/x for
 * native_recip).
 *
 * Pinning: tests/test_oracle_golden.py checks this restatement bit for bit
 * against fixtures in tests/golden/ that were produced by running the
 * reference's own kernel sources (compiled for x86 by oracle/Makefile into
 * oracle/_ref/, same built-in binding).  Beyond the reference's fixed
 * geometry (N=1024, 128 bins, 1024 waterfall rows, batch <= 1024) no
 * reference behaviour exists; this file then DEFINES the semantics by the
 * obvious generalisation (see fosphor_oracle.c) -- parity unpinned there.
 */
#ifndef FOSPHOR_ORACLE_H
#define FOSPHOR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fosphor_oracle fosphor_oracle;

/* Geometry.  Reference values: fft_len_log=10, n_bins=128, wf_rows=1024
 * (private.h:21-25, display.cl:96,163-165, cl.c:528,541). */
fosphor_oracle *fosphor_oracle_new(int fft_len_log, int n_bins, int wf_rows);
void            fosphor_oracle_free(fosphor_oracle *st);

/* fosphor.c:108-128 */
void fosphor_oracle_set_window_default(fosphor_oracle *st);
void fosphor_oracle_set_window(fosphor_oracle *st, const float *win);
/* fosphor.c:131-152 + cl.c:1081-1089 */
void fosphor_oracle_set_power_range(fosphor_oracle *st, int db_ref, int db_per_div);
/* kernel constants cl.c:714-716 (defaults 16, 1024, 0.002) */
void fosphor_oracle_set_constants(fosphor_oracle *st, float t0r, float t0d, float alpha);

/* cl.c:870-968.  len = complex samples.  Returns 0, or -EINVAL when len is
 * not a multiple of 16*N, or (strict != 0) exceeds 1024*N as the reference
 * host enforces (cl.c:882-886).  nthreads <= 1: single thread. */
int fosphor_oracle_process(fosphor_oracle *st, const float *samples, int len,
                           int strict, int nthreads);

/* Pieces, for kernel-level tests */
void fosphor_oracle_fft(int fft_len_log, const float *in, float *out,
                        const float *win, int n_spectra);
/* bin index of one FFT output sample, display.cl:136,161-168 */
int  fosphor_oracle_bin(float re, float im, float histo_scale, float histo_ofs, int n_bins);
void fosphor_oracle_twiddle(int radix2, int p, int k, int n, float *cs);
void fosphor_oracle_bins(const float *fft, int n, float histo_scale, float histo_ofs, int n_bins,
                         int32_t *bin, float *pwr);

/* State access (all row-major, layouts per cl.c:1003-1049 / private.h:40-42) */
float    *fosphor_oracle_waterfall(fosphor_oracle *st);   /* [wf_rows][N]        */
float    *fosphor_oracle_histogram(fosphor_oracle *st);   /* [n_bins][N]         */
float    *fosphor_oracle_spectrum(fosphor_oracle *st);    /* [2][N][2] live,max  */
uint32_t *fosphor_oracle_hitcount(fosphor_oracle *st);    /* [N][n_bins] last call */
float    *fosphor_oracle_fft_out(fosphor_oracle *st);     /* [batch][N][2] last call */
int       fosphor_oracle_waterfall_pos(fosphor_oracle *st);
float     fosphor_oracle_histo_scale(fosphor_oracle *st);
float     fosphor_oracle_histo_offset(fosphor_oracle *st);
const float *fosphor_oracle_window(fosphor_oracle *st);

#ifdef __cplusplus
}
#endif

#endif
