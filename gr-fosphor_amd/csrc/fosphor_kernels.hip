/*
 * fosphor_kernels.hip -- CDNA4 (gfx950) kernels of the fosphor compute core
 *
 * Replaces lib/fosphor/fft.cl + lib/fosphor/display.cl of the reference.
 * Compile with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
 * -ffp-contract=off is REQUIRED: the FFT must round exactly where the reference's
 * expressions round (mul, then add), or histogram counts stop being bit-exact.
 *
 * K1  k1_fft_bin   one WAVE per spectrum (64 lanes x 16 points), no block barriers.
 *                  Radix 8.8.8.2 Stockham with the reference's exact butterfly and
 *                  twiddle order (fft.cl:86-145,278-350,397-466); three exchanges
 *                  through an XOR-swizzled 8 KiB LDS slab per wave; twiddles and
 *                  window live in registers for the whole tile of spectra.
 *                  Epilogue per sample: |X|^2 -> v_log_f32 -> bin guess, accepted
 *                  when provably on the right side of a bin edge, otherwise decided
 *                  by comparing the double-precision |X|^2 with host-computed exact
 *                  thresholds (fosphor_portable_math.h) -- so integer bins equal the
 *                  oracle's log10(hypot()) pipeline bit for bit without evaluating it.
 * K2  k2_count     LDS-privatised histogram per (16-column slab, batch): ds_add on
 *                  [bin][col] (display.cl:161-177), plus the per-batch live sum / max.
 * K3  k3_merge     per (bin, x) cell rise/decay over all batches of the launch in
 *                  order (display.cl:217-254); live EMA + max-hold (display.cl:186-214,
 *                  257-310).
 */
#include "fosphor_internal.h"

#pragma clang fp contract(off)

namespace fosphor_amd {

/* Measurement switches that give WRONG RESULTS BY CONSTRUCTION (kernel parts skipped, chunks aliased) exist only in probe builds
 * (-DFOSPHOR_AMD_PROBES, tools/r04_ceiling_build.sh): in the product library they are the constant 0 and the code behind them is gone. */
#ifdef FOSPHOR_AMD_PROBES
#define PROBE_K1H(p)  ((p).dbg_k1h)
#define PROBE_SAME(p) ((p).dbg_same)
#else
#define PROBE_K1H(p)  0
#define PROBE_SAME(p) 0
#endif

/* ------------------------------------------------------------------------ */
/* Complex helpers: same operations, same order as fft.cl                   */
/* ------------------------------------------------------------------------ */
/* A complex value is one 64-bit VGPR pair (re, im).  Every helper performs exactly the IEEE
 * operations of the reference expression it cites -- only the instruction selection differs:
 * the half-swaps and sign flips of the reference's "multiply by -j" and of the
 * complex product ride on VOP3P op_sel / neg modifiers instead of costing v_mov / extra adds.
 * x - (-y) and x + y are the same IEEE operation, as are a*b and b*a, a+b and b+a.          */


typedef float v2f __attribute__((ext_vector_type(2)));

#define F_SQRT_1_2 (0.707106781188f)	/* fft.cl:72 */

/* fft.cl:37-46 : (a.x*w.x - a.y*w.y, a.x*w.y + a.y*w.x) */
static __device__ __forceinline__ v2f c_mul(v2f a, v2f w)
{
	v2f t1, t2, r;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(a), "v"(w));			/* (a.x*w.x, a.y*w.x) */
	asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t2) : "v"(a), "v"(w));	/* (a.y*w.y, a.x*w.y) */
	asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(t1), "v"(t2));			/* (t1.x - t2.x, t1.y + t2.y) */
	return r;
}
/* N complex products step by step (all first products, all second, all sums; pinned order): written one after the other, the dependent
 * statements of a product end up adjacent and the compiler puts an s_nop between them (it assumes a value written by inline assembly
 * cannot be forwarded) -- 28 issue slots per spectrum in the 1024-point kernel.  out[j] = in[j] * w[j], the same three operations. */
template <int N>
static __device__ __forceinline__ void c_mul_n(v2f (&out)[N], const v2f (&in)[N], const v2f (&w)[N])
{
	v2f t1[N], t2[N];
#pragma unroll
	for (int j = 0; j < N; j++)
		asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1[j]) : "v"(in[j]), "v"(w[j]));
#pragma unroll
	for (int j = 0; j < N; j++)
		asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t2[j]) : "v"(in[j]), "v"(w[j]));
#pragma unroll
	for (int j = 0; j < N; j++)
		asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(out[j]) : "v"(t1[j]), "v"(t2[j]));
}

/* a + mul_p1q2(b) and a - mul_p1q2(b), mul_p1q2(b) = (b.y, -b.x)  (fft.cl:77, used by dft8) */
static __device__ __forceinline__ v2f add_mj(v2f a, v2f b)
{
	v2f r;
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
static __device__ __forceinline__ v2f sub_mj(v2f a, v2f b)
{
	v2f r;
	asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

/* fft.cl:80 : SQRT_1_2 * (a.x + a.y, -a.x + a.y) */
static __device__ __forceinline__ v2f mul_p1q4(v2f a, v2f s12)
{
	v2f t, r;
	asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t) : "v"(a));	/* (a.x + a.y, a.y + -a.x) */
	asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(s12));
	return r;
}
/* fft.cl:82 : SQRT_1_2 * (-a.x + a.y, -a.x - a.y) */
static __device__ __forceinline__ v2f mul_p3q4(v2f a, v2f s12)
{
	v2f t, r;
	asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(t) : "v"(a));	/* (-a.x + a.y, -a.x + -a.y) */
	asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(s12));
	return r;
}

/* fft.cl:86-94 */
#define DFT2(a, b) do { v2f _t = (a) - (b); (a) = (a) + (b); (b) = _t; } while (0)
/* dft2(a, mul_p1q2(b)) */
#define DFT2_MJ(a, b) do { v2f _t = sub_mj((a), (b)); (a) = add_mj((a), (b)); (b) = _t; } while (0)

/* fft.cl:112-145.  The three mul_p1q2 twiddles (r6 after stage 1; r3, r7 after stage 2) are
 * folded into the butterflies that consume them. */
static __device__ __forceinline__ void dft8(v2f (&r)[8], v2f s12)
{
	DFT2(r[0], r[4]); DFT2(r[1], r[5]); DFT2(r[2], r[6]); DFT2(r[3], r[7]);
	r[5] = mul_p1q4(r[5], s12); r[7] = mul_p3q4(r[7], s12);
	DFT2(r[0], r[2]); DFT2(r[1], r[3]); DFT2_MJ(r[4], r[6]); DFT2(r[5], r[7]);
	DFT2(r[0], r[1]); DFT2_MJ(r[2], r[3]); DFT2(r[4], r[5]); DFT2_MJ(r[6], r[7]);
}

/* ------------------------------------------------------------------------ */
/* The long plans (N = 8192, 65536): twiddles on the butterflies, fused multiply-adds */
/* ------------------------------------------------------------------------ */
/* No reference behaviour exists at these lengths; the plan is this build's and the oracle restates it operation for operation
 * (oracle/fosphor_oracle.c: o_bf, o_bf_mj, o_bf_win, o_pass_radix16_fma, o_pass_radix2_fma -- the derivation is
 * written there).  A radix-R pass is log2 R radix-2 stages in decimation-in-time form, every butterfly
 *      a' = a + T b  (two v_pk_fma_f32)      b' = 2 a - a'  (one)
 * with T the twiddle of its stage and position: 36 packed operations per 8 points and pass instead of 49 (21 for seven complex
 * products + 28 for dft8), 96 per 16 points instead of 133, and 4 / 8 twiddles per item instead of 7 / 15.  Every operation is one
 * IEEE operation here and one fmaf / add / multiply there. */

/* o_bf: u = (a.x - b.y T.y, a.y + b.x T.y);  a' = (u.x + b.x T.x, u.y + b.y T.x);  b' = 2 a - a'
 * SC (the 8192-point kernel, which has no vector register to spare): `two` = (2, 2) travels in a scalar register pair -- a packed operation
 * takes one scalar source --, and so do the twiddles that are the same for every thread (bf_s, bf_mj_s: W8, W16, W16^3 of a first pass). */
template <bool SC = false>
static __device__ __forceinline__ void bf(v2f &a, v2f &b, v2f t, v2f two)
{
	v2f u, pa, nb;
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(u) : "v"(b), "v"(t), "v"(a));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(pa) : "v"(b), "v"(t), "v"(u));
	if (SC) asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb) : "v"(a), "s"(two), "v"(pa));
	else    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb) : "v"(a), "v"(two), "v"(pa));
	a = pa; b = nb;
}
/* o_bf_mj (T := -j T): u = (a.x + b.x T.y, a.y + b.y T.y);  a' = (u.x + b.y T.x, u.y - b.x T.x);  b' = 2 a - a' */
template <bool SC = false>
static __device__ __forceinline__ void bf_mj(v2f &a, v2f &b, v2f t, v2f two)
{
	v2f u, pa, nb;
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(u) : "v"(b), "v"(t), "v"(a));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(pa) : "v"(b), "v"(t), "v"(u));
	if (SC) asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb) : "v"(a), "s"(two), "v"(pa));
	else    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb) : "v"(a), "v"(two), "v"(pa));
	a = pa; b = nb;
}
/* ... with a twiddle that is the same for every thread */
template <bool SC = false>
static __device__ __forceinline__ void bf_s(v2f &a, v2f &b, v2f t, v2f two)
{
	if (!SC) { bf<false>(a, b, t, two); return; }
	v2f u, pa, nb;
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(u) : "v"(b), "s"(t), "v"(a));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(pa) : "v"(b), "s"(t), "v"(u));
	asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb) : "v"(a), "s"(two), "v"(pa));
	a = pa; b = nb;
}
template <bool SC = false>
static __device__ __forceinline__ void bf_mj_s(v2f &a, v2f &b, v2f t, v2f two)
{
	if (!SC) { bf_mj<false>(a, b, t, two); return; }
	v2f u, pa, nb;
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(u) : "v"(b), "s"(t), "v"(a));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(pa) : "v"(b), "s"(t), "v"(u));
	asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb) : "v"(a), "s"(two), "v"(pa));
	a = pa; b = nb;
}
/* Eight (CNT) butterflies of one stage, STEP BY STEP (all first operations, all second, all third): written butterfly by butterfly, dependent
 * inline-assembly statements end up adjacent and the compiler separates each such pair by an s_nop (it assumes a value written by inline
 * assembly cannot be forwarded): ~50 issue slots per thread and spectrum in the 8192-point kernel.  IA / IB: registers of the a / b inputs,
 * MJ: bit j set = butterfly j takes -j T (bf_mj). */
#ifndef BF_STAGEWISE
#define BF_STAGEWISE 1
#endif
template <bool SC, bool TS, int MJ, int I0, int I1, int I2, int I3, int I4, int I5, int I6, int I7, int D, int CNT = 8, bool SW = SC>
static __device__ __forceinline__ void bf8(v2f (&r)[16], v2f t0, v2f t1, v2f t2, v2f t3, v2f t4, v2f t5, v2f t6, v2f t7, v2f two)
{
	constexpr int ia[8] = { I0, I1, I2, I3, I4, I5, I6, I7 };
	const v2f t[8] = { t0, t1, t2, t3, t4, t5, t6, t7 };
	if (BF_STAGEWISE && SW) {		/* (SW: the 65536-point kernel measured 5 % slower in this form: 207 -> 217 registers under its skewed loop) */
	v2f u[8], pa[8], nb[8];
#pragma unroll
	for (int j = 0; j < CNT; j++) {
		if (MJ & (1 << j)) {
			if (TS) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(u[j]) : "v"(r[ia[j] + D]), "s"(t[j]), "v"(r[ia[j]]));
			else    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(u[j]) : "v"(r[ia[j] + D]), "v"(t[j]), "v"(r[ia[j]]));
		} else {
			if (TS) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(u[j]) : "v"(r[ia[j] + D]), "s"(t[j]), "v"(r[ia[j]]));
			else    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(u[j]) : "v"(r[ia[j] + D]), "v"(t[j]), "v"(r[ia[j]]));
		}
	}
#pragma unroll
	for (int j = 0; j < CNT; j++) {
		if (MJ & (1 << j)) {
			if (TS) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(pa[j]) : "v"(r[ia[j] + D]), "s"(t[j]), "v"(u[j]));
			else    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(pa[j]) : "v"(r[ia[j] + D]), "v"(t[j]), "v"(u[j]));
		} else {
			if (TS) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(pa[j]) : "v"(r[ia[j] + D]), "s"(t[j]), "v"(u[j]));
			else    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(pa[j]) : "v"(r[ia[j] + D]), "v"(t[j]), "v"(u[j]));
		}
	}
#pragma unroll
	for (int j = 0; j < CNT; j++) {
		if (SC) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb[j]) : "v"(r[ia[j]]), "s"(two), "v"(pa[j]));
		else    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(nb[j]) : "v"(r[ia[j]]), "v"(two), "v"(pa[j]));
	}
#pragma unroll
	for (int j = 0; j < CNT; j++) { r[ia[j]] = pa[j]; r[ia[j] + D] = nb[j]; }
	} else {
#pragma unroll
	for (int j = 0; j < CNT; j++) {
		if (MJ & (1 << j)) { if (TS) bf_mj_s<SC>(r[ia[j]], r[ia[j] + D], t[j], two); else bf_mj<SC>(r[ia[j]], r[ia[j] + D], t[j], two); }
		else               { if (TS) bf_s<SC>(r[ia[j]], r[ia[j] + D], t[j], two);    else bf<SC>(r[ia[j]], r[ia[j] + D], t[j], two); }
	}
	}
}

/* o_bf_win, stage A of the first pass: m = a wab.x;  a' = fma(b, wab.y, m);  b' = fma(-b, wab.y, m)   (wab = the two window taps) */
static __device__ __forceinline__ void bf_win(v2f &a, v2f &b, v2f wab)
{
	v2f m, pa, nb;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "v"(wab));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(pa) : "v"(b), "v"(wab), "v"(m));
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(nb) : "v"(b), "v"(wab), "v"(m));
	a = pa; b = nb;
}

/* o_pass_radix16_fma, p > 1, in two halves (the 65536-point kernel runs other work between them):
 * stages A, B: t8 = w^8, t4 = w^4;  stages C, D: t2 = w^2, t2w = w^2 W8, t1 = w, t1a = w W16, t1b = w W8, t1c = w W16^3.
 * X[m] is left in r[bitrev4(m)] (R16_PERM). */
template <bool SC = false, bool SW = SC>
static __device__ __forceinline__ void pass16_ab(v2f (&r)[16], v2f t8, v2f t4, v2f two)
{
	bf8<SC, false, 0x00, 0, 1, 2, 3, 4, 5, 6, 7, 8, 8, SW>(r, t8, t8, t8, t8, t8, t8, t8, t8, two);		/* stage A: (j, j + 8) */
	bf8<SC, false, 0xf0, 0, 1, 2, 3, 8, 9, 10, 11, 4, 8, SW>(r, t4, t4, t4, t4, t4, t4, t4, t4, two);		/* stage B: (j, j + 4); -j on the upper half */
}
template <bool SC = false, bool SW = SC>
static __device__ __forceinline__ void pass16_cd(v2f (&r)[16], v2f t2, v2f t2w, v2f t1, v2f t1a, v2f t1b, v2f t1c, v2f two)
{
	bf8<SC, false, 0xf0, 0, 1, 8, 9, 4, 5, 12, 13, 2, 8, SW>(r, t2, t2, t2w, t2w, t2, t2, t2w, t2w, two);	/* stage C: (j, j + 2) */
	bf8<SC, false, 0xaa, 0, 2, 4, 6, 8, 10, 12, 14, 1, 8, SW>(r, t1, t1, t1b, t1b, t1a, t1a, t1c, t1c, two);	/* stage D: (j, j + 1) */
}
/* ... p = 1: the window on stage A (wab[j] = taps of r[j], r[j + 8]); w16 = W16, w8 = W8, w163 = W16^3 */
template <bool SC = false, bool SW = SC>
static __device__ __forceinline__ void pass16_first(v2f (&r)[16], const v2f (&wab)[8], v2f w16, v2f w8, v2f w163, v2f two)
{
	if (BF_STAGEWISE && SW) {
		/* stage A step by step as well: the products, then the sums, then the differences */
		v2f m[8], pa[8], nb[8];
#pragma unroll
		for (int j = 0; j < 8; j++)
			asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(m[j]) : "v"(r[j]), "v"(wab[j]));
#pragma unroll
		for (int j = 0; j < 8; j++)
			asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(pa[j]) : "v"(r[j + 8]), "v"(wab[j]), "v"(m[j]));
#pragma unroll
		for (int j = 0; j < 8; j++)
			asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(nb[j]) : "v"(r[j + 8]), "v"(wab[j]), "v"(m[j]));
#pragma unroll
		for (int j = 0; j < 8; j++) { r[j] = pa[j]; r[j + 8] = nb[j]; }
	} else {
#pragma unroll
		for (int j = 0; j < 8; j++)
			bf_win(r[j], r[j + 8], wab[j]);
	}
#pragma unroll
	for (int j = 0; j < 4; j++) {
		DFT2(r[j], r[j + 4]);
		DFT2_MJ(r[8 + j], r[12 + j]);
	}
#pragma unroll
	for (int j = 0; j < 2; j++) {
		DFT2(r[j], r[j + 2]);
		DFT2_MJ(r[4 + j], r[6 + j]);
	}
	bf8<SC, true, 0x0c, 8, 9, 12, 13, 0, 0, 0, 0, 2, 4, SW>(r, w8, w8, w8, w8, w8, w8, w8, w8, two);	/* stage C, the twiddled half: (8, 10), (9, 11), -j: (12, 14), (13, 15) */
	DFT2(r[0], r[1]);          DFT2_MJ(r[2], r[3]);
	bf8<SC, true, 0x2a, 4, 6, 8, 10, 12, 14, 0, 0, 1, 6, SW>(r, w8, w8, w16, w16, w163, w163, w8, w8, two);	/* stage D: (4, 5) W8, -j (6, 7) W8, (8, 9) W16, -j (10, 11), (12, 13) W16^3, -j (14, 15) */
}

/* x * w with w broadcast from the low / high half of a pair (fft.cl:415-417) */
static __device__ __forceinline__ v2f mul_bcast_lo(v2f x, v2f w)
{
	v2f r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(w)); return r;
}
static __device__ __forceinline__ v2f mul_bcast_hi(v2f x, v2f w)
{
	v2f r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(w)); return r;
}

/* Order in which a radix-8 pass stores its outputs: offsets {0,p,..,7p} receive
 * r[0,4,2,6,1,5,3,7] (fft.cl:321-328). */
#define R8_PERM(jj) (((jj) == 0) ? 0 : ((jj) == 1) ? 4 : ((jj) == 2) ? 2 : ((jj) == 3) ? 6 : \
                     ((jj) == 4) ? 1 : ((jj) == 5) ? 5 : ((jj) == 6) ? 3 : 7)

/* Intra-wave LDS exchange: the store phase and the load phase of an exchange are separated by wavefront-scope release /
 * acquire fences around a wave barrier (the compiler must not move a load above a store it cannot prove aliases; the fences
 * cost an s_waitcnt lgkmcnt(0) each).  In principle program order alone would do -- all DS instructions of one wave execute in
 * order -- and a build with compiler-only barriers was measured in round 3: 519-525 against 523-526 GSamples/s, nothing; the
 * conservative form stays. */
static __device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ------------------------------------------------------------------------ */
/* Exact binning                                                            */
/* ------------------------------------------------------------------------ */

#define F_HALF_LOG10_2 (0.150514997831990597606869447362f)	/* pwr = log10|X| = this * log2(|X|^2) */
/* Inside K1 the log-power is carried as l2 = log2(|X|^2) and scaled once where it leaves. */

/* Decide a sample the fast path could not: compare |X|^2, formed in double with one
 * rounding (both squares are exact), against the exact thresholds.
 * thr[b] for b in [1, nb) = smallest double s with oracle_bin(s) >= b; thr[0] = -1;
 * thr[nb] = smallest s whose hypot overflows float (-> non-finite -> bin 0,
 * fosphor_portable_math.h fpm_bin_from_pwr). */
/* Where the exact path finds its thresholds.  It runs for ~1.5e-4 of the samples, i.e. in every sixth wave-spectrum, and a table load
 * through the vector memory path returns IN ORDER behind whatever the wave has in flight -- the next spectrum's IQ, requested from HBM
 * before the epilogue: measured on the 8192-point kernel, whose eight waves then all wait at the next barrier, 369 -> 318 us per launch
 * with the path removed.  Two ways around it:
 *   an LDS copy of the table (address_space(3) pointer: ds_read, its own counter) where the kernel has 2-4 KiB of LDS to spare;
 *   ThrScalar: the table entries fetched by the SCALAR unit (s_load_dwordx4 through the scalar cache, counted by lgkmcnt, out of order
 *   with the vector loads), one active lane after the other (usually there is exactly one). */
struct ThrScalar { const double *p; };
typedef uint32_t thr_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t thr_u2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ double thr_mk(uint32_t lo, uint32_t hi)
{
	return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
/* t0 = thr[i], t1 = thr[i + 1] for every active lane */
template <typename ThrPtr>
static __device__ __forceinline__ void thr_pair(ThrPtr thr, int i, double *t0, double *t1)
{
	*t0 = thr[i];
	*t1 = thr[i + 1];
}
template <>
__device__ __forceinline__ void thr_pair<ThrScalar>(ThrScalar thr, int i, double *t0, double *t1)
{
	double a = 0.0, b = 0.0;
	unsigned long long todo = __builtin_amdgcn_ballot_w64(true);		/* the active lanes */
	while (todo) {
		const int l = __builtin_ctzll(todo);
		const int g = __builtin_amdgcn_readlane(i, l);
		thr_u4 v;
		asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(thr.p + g) : "memory");
		const bool mine = (i == g);
		if (mine) { a = thr_mk(v.x, v.y); b = thr_mk(v.z, v.w); }
		todo &= ~__builtin_amdgcn_ballot_w64(mine);
	}
	*t0 = a; *t1 = b;
}
template <typename ThrPtr>
static __device__ __forceinline__ double thr_one(ThrPtr thr, int i)
{
	return thr[i];
}
template <>
__device__ __forceinline__ double thr_one<ThrScalar>(ThrScalar thr, int i)
{
	double a = 0.0;
	unsigned long long todo = __builtin_amdgcn_ballot_w64(true);
	while (todo) {
		const int l = __builtin_ctzll(todo);
		const int g = __builtin_amdgcn_readlane(i, l);
		thr_u2 v;
		asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(thr.p + g) : "memory");
		const bool mine = (i == g);
		if (mine) a = thr_mk(v.x, v.y);
		todo &= ~__builtin_amdgcn_ballot_w64(mine);
	}
	return a;
}

template <typename ThrPtr>		/* const double * (vector loads), an LDS pointer (a kernel's own copy of the table), or ThrScalar */
static __device__ __forceinline__ uint32_t bin_exact(float re, float im, float l2_fast, int guess,
                                                      ThrPtr thr, int nb, float *l2_out)
{
#ifdef K1_DBG_EXACT_BODY		/* measurement only: 1 = the branch with an empty body, 2 = the arithmetic without the table */
	if (K1_DBG_EXACT_BODY == 1) { *l2_out = l2_fast; asm volatile("s_nop 0"); return (uint32_t)guess; }
#endif
	const double xr = (double)re, xi = (double)im;
	const double sd = __builtin_fma(xr, xr, xi * xi);
	const float  sf = (float)sd;
	int bin;

#ifdef K1_DBG_EXACT_BODY
	if (K1_DBG_EXACT_BODY == 2) { *l2_out = l2_fast; return (uint32_t)(guess - (sd < 1.0 ? 1 : 0) + (sd >= 3.0 ? 1 : 0)) & 255u; }
#endif
	if (sf >= 1e-30f && sf <= 1e30f) {
		/* the guess is within one bin of the truth */
		double t0, t1;
		thr_pair(thr, guess, &t0, &t1);
		bin = guess - (sd < t0 ? 1 : 0) + (sd >= t1 ? 1 : 0);
		*l2_out = l2_fast;
	} else {
		/* zero, denormal, huge, inf or NaN: full search, and a log-power that does not
		 * depend on |X|^2 fitting a float: split sd = m * 2^e, m in [1,2) */
		int lo = 0, hi = nb;		/* invariant: sd >= thr[lo] (thr[0] = -1); sd < thr[hi] or hi == nb */
		const double t_top = thr_one(thr, nb);
		if (sd >= t_top) {
			bin = nb;
		} else if (!(sd >= 0.0)) {
			bin = 0;		/* NaN */
		} else {
			while (hi - lo > 1) {
				int mid = (lo + hi) >> 1;
				if (sd >= thr_one(thr, mid)) lo = mid; else hi = mid;
			}
			bin = lo;
		}
		if (__builtin_isinf(re) || __builtin_isinf(im) || sd >= t_top) {
			*l2_out = __builtin_inff();		/* hypot(inf, anything) = inf; float hypot overflow */
		} else if (sd == 0.0) {
			*l2_out = -__builtin_inff();		/* log10(0) */
		} else if (sd != sd) {
			*l2_out = __builtin_nanf("");
		} else {
			const unsigned long long u = (unsigned long long)__double_as_longlong(sd);
			const int e = (int)((u >> 52) & 0x7ff) - 1023;
			const double m = __longlong_as_double((long long)((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
			*l2_out = (float)e + __builtin_amdgcn_logf((float)m);
		}
	}
	if (bin >= nb)
		bin = 0;
	return (uint32_t)bin;
}

struct BinConst { float A, C, amb, kappa; int nb; const double *thr; };

/* Fast path.  Returns r = rint(v) (the bin guess before clamping, as a float), l2 = log2(|X|^2)
 * and the sample's ambiguity measure
 *     amb = |v - r| + kappa * |l2|
 * -- distance of the scaled log-power from the bin centre, plus the v_log_f32 error bound
 * (<= 1 ulp of l2, through the slope A: kappa = 2 * A * 2^-23, the factor 2 is margin; it also
 * covers the rounding of s32).  The guess is provably exact iff amb <= 0.5 - delta0, delta0
 * bounding the roundings that do not scale with l2.  amb is >= 0,
 * +inf for |X|^2 in {0, denormal-flushed, inf}, NaN for NaN: compared as an unsigned bit
 * pattern all of those order above every finite value, so one running v_max_u32 per spectrum
 * collects "some sample needs the exact path" without per-sample compares or branches. */
#ifndef K1_DBG_EPI
#define K1_DBG_EPI 0		/* measurement only (wrong results): 1 no v_log_f32, 2 no ambiguity measure, 4 no live / max update, 8 no bin byte,
				 * 16 no bin-index stores, 32 sixteen LDS atomics per spectrum on a dummy counter array (what counting inside K1
				 * would issue): the probe builds of profiles/r04_ceiling.md (tools/r04_ceiling_build.sh) */
#endif
#ifndef K1_LATE_BINS
#define K1_LATE_BINS 1			/* 0: the bin dwords of a quad stored where the quad ends (A/B builds) */
#endif
#ifndef K1_THR_LDS
#define K1_THR_LDS 1			/* 0: the N = 1024 kernel reads the thresholds from memory (A/B builds) */
#endif
#ifndef K1_DBG_NO_EXACT
#define K1_DBG_NO_EXACT 0		/* measurement only: 1 drops the exact path (wrong bins on near-ties) */
#endif
static __device__ __forceinline__ float bin_fast(float re, float im, const BinConst &k, float *l2_out, uint32_t *amb_bits)
{
	const float s  = __builtin_fmaf(re, re, im * im);
	const float l2 = (K1_DBG_EPI & 1) ? s : __builtin_amdgcn_logf(s);		/* v_log_f32 */
	const float v  = __builtin_fmaf(k.A, l2, k.C);
	const float r  = __builtin_rintf(v);
	const float a  = (K1_DBG_EPI & 2) ? 0.0f : __builtin_fmaf(__builtin_fabsf(l2), k.kappa, __builtin_fabsf(v - r));
	*l2_out = l2;
	*amb_bits = __float_as_uint(a);
	return r;
}

/* bin byte of a float guess r: saturating float -> u8 (NaN -> 0), capped at nb-1 */
static __device__ __forceinline__ uint32_t pack_bin(float r, float top, uint32_t byte, uint32_t old)
{
	return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fminf(r, top), byte, old);
}

static __device__ __forceinline__ float max_f32(float a, float b)
{
	/* one v_max_f32 (IEEE mode: a NaN operand is dropped = OpenCL max(acc, NaN) keeps acc,
	 * display.cl:139); fmaxf() would add two canonicalising self-max instructions */
	float r;
	asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

/* ------------------------------------------------------------------------ */
/* K1                                                                       */
/* ------------------------------------------------------------------------ */

/* Tunables (tools/ab_bench.sh builds variants with -D...) */
#ifndef K1_WAVES_PER_SIMD
#define K1_WAVES_PER_SIMD 2		/* __launch_bounds__ second argument */
#endif

/* K1_TIMING=1 (debug builds only, tools/k1_phase_timing.py): s_memtime stamps per phase,
 * accumulated per wave into K1Params::dbg[wave][phase]. */
#ifndef K1_TIMING
#define K1_TIMING 0
#endif
#if K1_TIMING
#define K1_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
		const long long _now = __builtin_readcyclecounter(); tacc[i] += _now - tprev; tprev = _now; } while (0)
#else
#define K1_STAMP(i) do { } while (0)
#endif

typedef float v4f __attribute__((ext_vector_type(4)));

/* 8 x (64 lanes x 16 B) = 1 KiB per instruction, read-once: non-temporal.  `src` points at this
 * lane's pair: elements (2L, 2L+1) + 128k land in x[2k], x[2k+1].  (Ablation on MI355X: with
 * 8-byte-per-lane loads the load path alone caps K1 near 4.7 TB/s; 16-byte ones do not.) */
static __device__ __forceinline__ void load_iq16(v2f (&x)[16], const float2 *__restrict__ src)
{
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src + 128 * k));
		x[2 * k]     = v2f{ q.x, q.y };
		x[2 * k + 1] = v2f{ q.z, q.w };
	}
}
#define K1_LANE_SRC(lane) (2 * (lane))

template <bool WRITE_FFT, bool NB256 = false>	/* NB256: 256 bins -- the saturating conversion of the bin byte IS the clamp at n_bins - 1 */
__global__ __launch_bounds__(256, K1_WAVES_PER_SIMD)
void k1_fft_bin(const K1Params p)
{
	__shared__ v2f   lds[4][kN];			/* 8 KiB exchange slab per wave */
	__shared__ v2f   tw4_tab[512];			/* pass-4 twiddles, shared by the block */
	__shared__ float win_tab[kN];			/* window, shared by the block */
	/* exact-bin thresholds (n_bins <= 256 in this kernel): the rare path that consults them would otherwise wait for its two table
	 * loads BEHIND the next spectrum's IQ, already requested from HBM -- loads return in order */
	__shared__ double thr_tab[264];
#if K1_DBG_EPI & 32
	__shared__ uint32_t dbg_cnt[256 * 32];		/* probe: the counter image of a 64-column slab */
#endif

	const int lane   = threadIdx.x & 63;
	const int wv     = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);	/* tile, spectrum index, row predicate: SGPRs */
	const int ntiles = p.total / p.tile;
	const int stride = gridDim.x * 4;		/* waves in the grid */
#if K1_TIMING
	const long long t_wave_start = wall_clock64();	/* 100 MHz, common to all CUs */
#endif
	int tile = blockIdx.x * 4 + wv;
	const v2f *twg = reinterpret_cast<const v2f *>(p.tw);

	for (int i = threadIdx.x; i < kN; i += 256)
		win_tab[i] = p.win[i];
	for (int i = threadIdx.x; i < 512; i += 256)
		tw4_tab[i] = twg[kTw4Off + i];
	for (int i = threadIdx.x; i <= p.n_bins && i < 264; i += 256)
		thr_tab[i] = p.thr[i];
	__syncthreads();				/* the only block-wide barrier */

	if (tile >= ntiles)
		return;					/* whole wave leaves */

	v2f *buf = lds[wv];

	/* ---- per-lane constants, loaded once per wave -------------------------- */
	v2f tw2[7];
	v2f tw3[7];
#pragma unroll
	for (int n = 0; n < 7; n++) {
		tw2[n] = twg[kTw2Off + (lane & 7) * 7 + n];	/* k = i & 7  (both virtual items) */
		tw3[n] = twg[kTw3Off + lane * 7 + n];		/* k = i & 63 = lane               */
	}
	const v2f s12 = { F_SQRT_1_2, F_SQRT_1_2 };

	/* ---- swizzled LDS addressing -------------------------------------------
	 * element e lives at phys(e) = e ^ ((e >> 3) & 15): every access below is
	 * bank-conflict free for ds_read_b64 (32-lane groups, 64 banks) and
	 * ds_write_b64 (16-lane groups, 32 banks).  The closed forms per access
	 * pattern are derived in DESIGN_HISTORY.md ("LDS exchange").                    */
	const int rd_even = lane ^ ((lane >> 3) & 7);		/* e = lane + 64m, m even */
	const int rd_odd  = rd_even ^ 8;			/*                 m odd  */
	const int st1     = (8 * lane) ^ (lane & 15);		/* pass 1: e = 8i + jj, i = lane (+64v)   */
	const int st1a    = (16 * lane) ^ ((2 * lane) & 15);		/* pass 1, i = 2 lane     */
	const int st1b    = (16 * lane + 8) ^ ((2 * lane + 1) & 15);	/* pass 1, i = 2 lane + 1 */
	(void)st1; (void)st1a; (void)st1b;
	const int st2     = ((64 * (lane >> 3)) + (lane & 7)) ^ (lane & 8);	/* pass 2: e = 64(i>>3)+(i&7)+8jj */

	const BinConst bk = { p.binA, p.binC, p.amb, p.kappa, p.n_bins, p.thr };
	const float vmax_init = -1000.0f / F_HALF_LOG10_2;	/* display.cl:91, in log2 units */

	v2f xn[16];
	load_iq16(xn, p.iq + (size_t)tile * p.tile * p.hop + K1_LANE_SRC(lane));
#if K1_TIMING
	long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	long long tprev = __builtin_readcyclecounter();
#endif

	/* persistent wave: tiles tile, tile + stride, ... (per-lane constants stay in registers) */
	for (; tile < ntiles; tile += stride) {
	const int t0 = tile * p.tile;

	/* live partial and running max of this tile, in log2(|X|^2) units */
	float live[16], vmax[16];
#pragma unroll
	for (int m = 0; m < 16; m++) {
		live[m] = 0.0f;
		vmax[m] = vmax_init;
	}

	/* The bin dwords of a quad of spectra are stored one window multiply LATER than they are complete: the wait for the prefetched IQ at
	 * the top of a spectrum is an s_waitcnt vmcnt(0) (the number of stores behind the loads varies, so the compiler cannot count them out),
	 * and stores issued behind those loads -- at the end of the previous spectrum -- made every fourth spectrum wait for its own stores'
	 * acknowledgements.  Stores issued AHEAD of the next prefetch are older than the loads the wave waits for next. */
	uint32_t pack[16];
#pragma unroll
	for (int m = 0; m < 16; m++)
		pack[m] = 0;
	int pend_row = -1;			/* row of p.bins the bytes in pack belong to, or -1 (uniform) */
	auto flush_pack = [&]() {
		uint32_t *dst = p.bins + (size_t)pend_row * kN + lane;
		if (K1_DBG_EPI & 16) {
			uint32_t any = 0;			/* keep the values alive without the stores */
#pragma unroll
			for (int m = 0; m < 16; m++)
				any |= pack[m];
			if (any == 0xdeadbeefu)
				dst[0] = any;
		} else {
#pragma unroll
			for (int m = 0; m < 16; m++)
				dst[64 * m] = pack[m];
		}
#pragma unroll
		for (int m = 0; m < 16; m++)
			pack[m] = 0;
		pend_row = -1;
	};

	for (int g0 = 0; g0 < p.tile; g0 += 4) {
#pragma unroll 1
		for (int u = 0; u < 4; u++) {
			const int t = t0 + g0 + u;
			v2f x[16];

			K1_STAMP(7);		/* loop overhead + stores of the previous iteration */
			/* window (fft.cl:415-417); taps fetched as pairs */
#pragma unroll
			for (int k = 0; k < 8; k++) {	/* x[2k], x[2k+1] = elements 2L + 128k, 2L + 1 + 128k */
				const v2f w = *reinterpret_cast<const v2f *>(&win_tab[2 * lane + 128 * k]);
				x[2 * k]     = mul_bcast_lo(xn[2 * k], w);
				x[2 * k + 1] = mul_bcast_hi(xn[2 * k + 1], w);
			}

			if (K1_LATE_BINS && u == 0 && pend_row >= 0)
				flush_pack();		/* the previous quad's bin dwords: behind the wait above, ahead of the prefetch below */
			/* prefetch the next spectrum this wave will process */
			{
				const bool last = (g0 + u + 1 == p.tile);
				const int t_next = last ? (tile + stride) * p.tile : t + 1;
				if (!last || tile + stride < ntiles)
					load_iq16(xn, p.iq + (size_t)t_next * p.hop + K1_LANE_SRC(lane));
			}

			K1_STAMP(0);		/* window (includes waiting for the prefetched IQ) + prefetch issue */
			/* ---- pass 1: radix 8, p = 1, no twiddle (fft.cl:419-420) --------
			 * This lane is virtual work-items i = 2L + v (elements i + 128j = x[2j + v], as the 16-byte
			 * loads deliver them).  Item i stores its outputs at e = 8i + jj; which lane runs which
			 * item is free. */
#pragma unroll
			for (int v = 0; v < 2; v++) {
				v2f r[8];
#pragma unroll
				for (int j = 0; j < 8; j++)
					r[j] = x[v + 2 * j];
				dft8(r, s12);
#pragma unroll
				for (int jj = 0; jj < 8; jj++)
					buf[(v ? st1b : st1a) ^ jj] = r[R8_PERM(jj)];
			}
			wave_lds_sync();
#pragma unroll
			for (int m = 0; m < 16; m++)
				x[m] = buf[((m & 1) ? rd_odd : rd_even) + 64 * m];
			wave_lds_sync();

			K1_STAMP(1);		/* pass 1 + exchange */
			/* ---- pass 2: radix 8, p = 8 (fft.cl:422-423) ------------------- */
#pragma unroll
			for (int v = 0; v < 2; v++) {
				v2f r[8];
				{
					v2f in7[7], out7[7];
#pragma unroll
					for (int j = 1; j < 8; j++)
						in7[j - 1] = x[v + 2 * j];
					c_mul_n<7>(out7, in7, tw2);
					r[0] = x[v];
#pragma unroll
					for (int j = 1; j < 8; j++)
						r[j] = out7[j - 1];
				}
				dft8(r, s12);
#pragma unroll
				for (int jj = 0; jj < 8; jj++)
					buf[(st2 ^ (9 * jj)) + 512 * v] = r[R8_PERM(jj)];
			}
			wave_lds_sync();
#pragma unroll
			for (int m = 0; m < 16; m++)
				x[m] = buf[((m & 1) ? rd_odd : rd_even) + 64 * m];
			wave_lds_sync();

			K1_STAMP(2);		/* pass 2 + exchange */
			/* ---- pass 3: radix 8, p = 64 (fft.cl:425-426) ------------------
			 * Virtual item i = lane + 64v stores its outputs at e = 512v + lane + 64jj, and the
			 * pass-4 butterflies of this lane read exactly e = lane + 64m: with both items of a
			 * pair in the same lane the third exchange is the identity x[jj + 8v] = out_v[jj] --
			 * no LDS round trip (fft.cl:347-349 + 435-438 collapse to register renaming). */
			{
				v2f y[16];
#pragma unroll
				for (int v = 0; v < 2; v++) {
					v2f r[8];
					{
						v2f in7[7], out7[7];
#pragma unroll
						for (int j = 1; j < 8; j++)
							in7[j - 1] = x[v + 2 * j];
						c_mul_n<7>(out7, in7, tw3);
						r[0] = x[v];
#pragma unroll
						for (int j = 1; j < 8; j++)
							r[j] = out7[j - 1];
					}
					dft8(r, s12);
#pragma unroll
					for (int jj = 0; jj < 8; jj++)
						y[jj + 8 * v] = r[R8_PERM(jj)];
				}
#pragma unroll
				for (int m = 0; m < 16; m++)
					x[m] = y[m];
			}

			K1_STAMP(3);		/* pass 3 + exchange */
			/* ---- pass 4: radix 2, p = 512 (fft.cl:428-458) ------------------
			 * butterfly on elements (j, j + 512), j = lane + 64c, twiddle k = j.
			 * Results: X[j] -> x[c], X[j + 512] -> x[c + 8], i.e. column lane + 64m. */
			{
				v2f in8[8], w8[8], out8[8];
#pragma unroll
				for (int c = 0; c < 8; c++) { in8[c] = x[c + 8]; w8[c] = tw4_tab[lane + 64 * c]; }	/* k = lane + 64c */
				c_mul_n<8>(out8, in8, w8);
#pragma unroll
				for (int c = 0; c < 8; c++) {
					v2f a = x[c];
					v2f b = out8[c];
					DFT2(a, b);
					x[c] = a;
					x[c + 8] = b;
				}
			}

			K1_STAMP(4);		/* pass 4 */
			if (WRITE_FFT) {
#pragma unroll
				for (int m = 0; m < 16; m++)
					reinterpret_cast<v2f *>(p.fft_out)[(size_t)t * kN + lane + 64 * m] = x[m];
			}

			/* ---- epilogue: log-power, exact bin (display.cl:136,161-168) ---- */
			float    l2[16];
			uint32_t amb = 0;
			const float top = (float)(bk.nb - 1);
#pragma unroll
			for (int m = 0; m < 16; m++) {
				uint32_t ab;
				const float r = bin_fast(x[m].x, x[m].y, bk, &l2[m], &ab);
				amb = amb > ab ? amb : ab;			/* v_max_u32: NaN / inf propagate */
				if (!(K1_DBG_EPI & 8))
					pack[m] = NB256 ? __builtin_amdgcn_cvt_pk_u8_f32(r, (uint32_t)u, pack[m]) : pack_bin(r, top, (uint32_t)u, pack[m]);
				else
					pack[m] ^= __float_as_uint(r);
			}
			if (!K1_DBG_NO_EXACT && amb > __float_as_uint(bk.amb)) {
				/* rare (a few % of spectra have one such sample): find the samples, decide them
				 * against the exact thresholds, patch their bin byte and log-power */
#pragma unroll
				for (int m = 0; m < 16; m++) {
					const float v = __builtin_fmaf(bk.A, l2[m], bk.C);
					const float r = __builtin_rintf(v);
					const float a = __builtin_fmaf(__builtin_fabsf(l2[m]), bk.kappa, __builtin_fabsf(v - r));
					if (!(a <= bk.amb)) {
						const int guess = (int)__builtin_amdgcn_fmed3f(r, 0.0f, top);
						float nl2;
#if K1_THR_LDS
						const uint32_t nbn = bin_exact(x[m].x, x[m].y, l2[m], guess,
						                               (const __attribute__((address_space(3))) double *)thr_tab, bk.nb, &nl2);
#else
						const uint32_t nbn = bin_exact(x[m].x, x[m].y, l2[m], guess, bk.thr, bk.nb, &nl2);
#endif
						pack[m] = (pack[m] & ~(0xffu << (8 * u))) | (nbn << (8 * u));
						l2[m] = nl2;
					}
				}
			}

#pragma unroll
			for (int m = 0; m < 16; m++) {
				/* Horner form of display.cl:149-150, in place (v_fma with the accumulator as destination:
				 * the compiler's v_fmac into the dying l2 register costs a v_mov per column) */
				if (K1_DBG_EPI & 4) { live[m] = l2[m]; continue; }
				asm("v_fma_f32 %0, %0, %1, %2" : "+v"(live[m]) : "s"(p.w), "v"(l2[m]));
				vmax[m] = max_f32(vmax[m], l2[m]);		/* display.cl:139 */
			}
#if K1_DBG_EPI & 32
#pragma unroll
			for (int m = 0; m < 16; m++)
				atomicAdd(&dbg_cnt[((pack[m] >> (8 * u)) & 0xffu) * 32 + (lane & 31)], (lane & 32) ? 0x10000u : 1u);
#endif
			if (t >= p.wf_first) {				/* uniform: one scalar branch */
				float *wf_row = p.wf + (size_t)((p.wf_pos0 + t) & p.wf_mask) * kN + lane;
#pragma unroll
				for (int m = 0; m < 16; m++)
					wf_row[64 * m] = l2[m] * F_HALF_LOG10_2;	/* display.cl:142-146 */
			}
			K1_STAMP(6);		/* epilogue */
		}

		K1_STAMP(5);			/* 4th epilogue (the first three land in 7) */
		/* 4 spectra x 1 column per dword, coalesced 256 B per instruction: stored at the top of the next quad (or below) */
		pend_row = (t0 + g0) >> 2;
		if (!K1_LATE_BINS)
			flush_pack();
	}
	if (pend_row >= 0)
		flush_pack();

	/* leave the log2 domain: pwr = log10|X| = l2 * log10(2)/2; an untouched max is exactly -1000 */
	float2 *pp = p.partial + (size_t)tile * kN + lane;
#pragma unroll
	for (int m = 0; m < 16; m++)
		pp[64 * m] = make_float2(live[m] * F_HALF_LOG10_2,
		                         (vmax[m] == vmax_init) ? -1000.0f : vmax[m] * F_HALF_LOG10_2);
	}	/* tile loop */
#if K1_TIMING
	if (p.dbg && lane == 0) {
		const int w = blockIdx.x * 4 + wv;
		for (int i = 0; i < 8; i++)
			p.dbg[w * 8 + i] = tacc[i];
		/* wave lifetime on the common clock replaces the two near-empty phase slots */
		p.dbg[w * 8 + 3] = t_wave_start;
		p.dbg[w * 8 + 5] = wall_clock64();
	}
#endif
}

/* ------------------------------------------------------------------------ */
/* K1's memory traffic without K1's arithmetic (measurement hook)             */
/* ------------------------------------------------------------------------ */
/* The same persistent grid, tile order, 16-byte non-temporal loads one spectrum ahead, and the same
 * stores (bin dwords every 4 spectra, tile partials every tile) as k1_fft_bin -- and nothing else.
 * Its duration is the practical floor the memory system sets for K1 on this chip: bench.py reports
 * K1's duration next to it (roofline.traffic_twin). */
__global__ __launch_bounds__(256, K1_WAVES_PER_SIMD)
void k1_traffic_twin(const K1Params p)
{
	const int lane   = threadIdx.x & 63;
	const int wv     = threadIdx.x >> 6;
	const int ntiles = p.total / p.tile;
	const int stride = gridDim.x * 4;
	int tile = blockIdx.x * 4 + wv;
	if (tile >= ntiles)
		return;
	v2f xn[16];
	load_iq16(xn, p.iq + (size_t)tile * p.tile * p.hop + K1_LANE_SRC(lane));
	for (; tile < ntiles; tile += stride) {
		const int t0 = tile * p.tile;
		v2f acc = { 0.0f, 0.0f };
		for (int g0 = 0; g0 < p.tile; g0 += 4) {
#pragma unroll 1
			for (int u = 0; u < 4; u++) {
				const int t = t0 + g0 + u;
				v2f x[16];
#pragma unroll
				for (int m = 0; m < 16; m++)
					x[m] = xn[m];
				const bool last = (g0 + u + 1 == p.tile);
				const int t_next = last ? (tile + stride) * p.tile : t + 1;
				if (!last || tile + stride < ntiles)
					load_iq16(xn, p.iq + (size_t)t_next * p.hop + K1_LANE_SRC(lane));
#pragma unroll
				for (int m = 0; m < 16; m++)
					acc += x[m];			/* consume the data: 16 adds per spectrum */
			}
			uint32_t *dst = p.bins + (size_t)((t0 + g0) >> 2) * kN + lane;
#pragma unroll
			for (int m = 0; m < 16; m++)
				dst[64 * m] = __float_as_uint(acc.x) + (uint32_t)m;
		}
		float2 *pp = p.partial + (size_t)tile * kN + lane;
#pragma unroll
		for (int m = 0; m < 16; m++)
			pp[64 * m] = make_float2(acc.x, acc.y + (float)m);
	}
}

hipError_t launch_k1_traffic_twin(const K1Params &p, hipStream_t s)
{
	const int tiles = p.total / p.tile;
	int blocks = (tiles + 3) / 4;
	if (blocks > kK1MaxBlocks)
		blocks = kK1MaxBlocks;
	hipLaunchKernelGGL(k1_traffic_twin, dim3(blocks), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* K1 v2: two waves per spectrum                                             */
/* ------------------------------------------------------------------------ */
/* Same arithmetic, same LDS layout, same outputs as k1_fft_bin, but a spectrum is shared by
 * the two waves of a 128-thread work-group exactly like the reference's 128 work-items
 * (fft.cl:403: WG_SIZE = N/8): lane l of wave w IS virtual work-item i = l + 64w and owns 8
 * points.  Every per-lane array halves (x, prefetch, l2, live/max, pack), which fits
 * 4 waves per SIMD without spills; the price is one 2-wave s_barrier per exchange.
 * After pass 3 wave w takes the pass-4 butterflies c in [4w, 4w+4), i.e. columns
 * lane + 64m for m in {4w..4w+3} U {8+4w..8+4w+3}. */
#ifndef K1V2_WAVES_PER_SIMD
#define K1V2_WAVES_PER_SIMD 3
#endif

static __device__ __forceinline__ void load_iq8(v2f (&x)[8], const float2 *__restrict__ src)
{
#pragma unroll
	for (int j = 0; j < 8; j++)		/* elements i + 128 j */
		x[j] = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(src + 128 * j));
}

template <bool WRITE_FFT>
__global__ __launch_bounds__(128, K1V2_WAVES_PER_SIMD)
void k1v2_fft_bin(const K1Params p)
{
	__shared__ v2f   buf[kN];			/* 8 KiB exchange slab of the work-group's spectrum */
	__shared__ v2f   tw4_tab[512];
	__shared__ float win_tab[kN];

	const int lane   = threadIdx.x & 63;
	const int w      = threadIdx.x >> 6;		/* wave = virtual-item half */
	const int i0     = threadIdx.x;			/* virtual work-item i = lane + 64w */
	const int ntiles = p.total / p.tile;
	const int stride = gridDim.x;
	int tile = blockIdx.x;
	const v2f *twg = reinterpret_cast<const v2f *>(p.tw);

	for (int i = threadIdx.x; i < kN; i += 128)
		win_tab[i] = p.win[i];
	for (int i = threadIdx.x; i < 512; i += 128)
		tw4_tab[i] = twg[kTw4Off + i];
	__syncthreads();

	/* per-lane twiddles: k = i & 7 and k = i & 63 do not depend on w */
	v2f tw2[7];
	v2f tw3[7];
#pragma unroll
	for (int n = 0; n < 7; n++) {
		tw2[n] = twg[kTw2Off + (lane & 7) * 7 + n];
		tw3[n] = twg[kTw3Off + lane * 7 + n];
	}
	const v2f s12 = { F_SQRT_1_2, F_SQRT_1_2 };

	/* swizzled addressing, as in k1_fft_bin with v = w */
	const int rd_even = lane ^ ((lane >> 3) & 7);
	const int rd_w    = w ? (rd_even ^ 8) : rd_even;		/* e = lane + 64(w + 2j): parity of m is w */
	const int st1     = ((8 * lane) ^ (lane & 15)) + 512 * w;
	const int st2     = (((64 * (lane >> 3)) + (lane & 7)) ^ (lane & 8)) + 512 * w;
	const int st3     = 512 * w;					/* + (odd jj ? rd_odd : rd_even) + 64 jj */
	const int rd_odd  = rd_even ^ 8;

	const BinConst bk = { p.binA, p.binC, p.amb, p.kappa, p.n_bins, p.thr };
	const float vmax_init = -1000.0f / F_HALF_LOG10_2;
	const float top = (float)(bk.nb - 1);

	v2f xn[8];
	if (tile < ntiles)
		load_iq8(xn, p.iq + (size_t)tile * p.tile * p.hop + i0);

	for (; tile < ntiles; tile += stride) {		/* uniform over the work-group */
	const int t0 = tile * p.tile;

	float live[8], vmax[8];
#pragma unroll
	for (int q = 0; q < 8; q++) {
		live[q] = 0.0f;
		vmax[q] = vmax_init;
	}

	for (int g0 = 0; g0 < p.tile; g0 += 4) {
		uint32_t pack[8];
#pragma unroll
		for (int q = 0; q < 8; q++)
			pack[q] = 0;

#pragma unroll 1
		for (int u = 0; u < 4; u++) {
			const int t = t0 + g0 + u;
			v2f r[8];

			/* window (fft.cl:415-417) */
#pragma unroll
			for (int j = 0; j < 8; j += 2) {
				v2f ww;
				ww.x = win_tab[i0 + 128 * j];
				ww.y = win_tab[i0 + 128 * (j + 1)];
				r[j]     = mul_bcast_lo(xn[j], ww);
				r[j + 1] = mul_bcast_hi(xn[j + 1], ww);
			}
			{	/* prefetch the next spectrum of this work-group */
				const bool last = (g0 + u + 1 == p.tile);
				const int t_next = last ? (tile + stride) * p.tile : t + 1;
				if (!last || tile + stride < ntiles)
					load_iq8(xn, p.iq + (size_t)t_next * p.hop + i0);
			}

			/* pass 1 (fft.cl:419-420) */
			dft8(r, s12);
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
				buf[st1 ^ jj] = r[R8_PERM(jj)];
			__syncthreads();
#pragma unroll
			for (int j = 0; j < 8; j++)
				r[j] = buf[rd_w + 64 * (w + 2 * j)];
			__syncthreads();

			/* pass 2 (fft.cl:422-423) */
#pragma unroll
			for (int j = 1; j < 8; j++)
				r[j] = c_mul(r[j], tw2[j - 1]);
			dft8(r, s12);
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
				buf[st2 ^ (9 * jj)] = r[R8_PERM(jj)];
			__syncthreads();
#pragma unroll
			for (int j = 0; j < 8; j++)
				r[j] = buf[rd_w + 64 * (w + 2 * j)];
			__syncthreads();

			/* pass 3 (fft.cl:425-426) */
#pragma unroll
			for (int j = 1; j < 8; j++)
				r[j] = c_mul(r[j], tw3[j - 1]);
			dft8(r, s12);
#pragma unroll
			for (int jj = 0; jj < 8; jj++)
				buf[st3 + ((jj & 1) ? rd_odd : rd_even) + 64 * jj] = r[R8_PERM(jj)];
			__syncthreads();

			/* pass 4 (fft.cl:428-458): butterflies c = 4w + q on elements (j, j+512), j = lane + 64c.
			 * x[q] = X[lane + 64(4w+q)], x[q+4] = X[lane + 64(8+4w+q)] */
			v2f x[8];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int c = 4 * w + q;		/* parity of c is parity of q */
				v2f a = buf[((q & 1) ? rd_odd : rd_even) + 64 * c];
				v2f b = buf[((q & 1) ? rd_odd : rd_even) + 64 * (c + 8)];
				b = c_mul(b, tw4_tab[lane + 64 * c]);
				DFT2(a, b);
				x[q] = a;
				x[q + 4] = b;
			}
			__syncthreads();		/* the slab is rewritten by the next spectrum's pass 1 */

			if (WRITE_FFT) {
#pragma unroll
				for (int q = 0; q < 4; q++) {
					reinterpret_cast<v2f *>(p.fft_out)[(size_t)t * kN + lane + 64 * (4 * w + q)] = x[q];
					reinterpret_cast<v2f *>(p.fft_out)[(size_t)t * kN + lane + 64 * (8 + 4 * w + q)] = x[q + 4];
				}
			}

			/* epilogue (display.cl:136,161-168), as in k1_fft_bin */
			float    l2[8];
			uint32_t amb = 0;
#pragma unroll
			for (int q = 0; q < 8; q++) {
				uint32_t ab;
				const float rr = bin_fast(x[q].x, x[q].y, bk, &l2[q], &ab);
				amb = amb > ab ? amb : ab;
				pack[q] = pack_bin(rr, top, (uint32_t)u, pack[q]);
			}
			if (amb > __float_as_uint(bk.amb)) {
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const float v = __builtin_fmaf(bk.A, l2[q], bk.C);
					const float rr = __builtin_rintf(v);
					const float a = __builtin_fmaf(__builtin_fabsf(l2[q]), bk.kappa, __builtin_fabsf(v - rr));
					if (!(a <= bk.amb)) {
						const int guess = (int)__builtin_amdgcn_fmed3f(rr, 0.0f, top);
						float nl2;
						const uint32_t nbn = bin_exact(x[q].x, x[q].y, l2[q], guess, ThrScalar{ bk.thr }, bk.nb, &nl2);
						pack[q] = (pack[q] & ~(0xffu << (8 * u))) | (nbn << (8 * u));
						l2[q] = nl2;
					}
				}
			}

			const bool store_row = (t >= p.wf_first);
			float *wf_row = p.wf + (size_t)((p.wf_pos0 + t) & p.wf_mask) * kN + lane + 256 * w;
#pragma unroll
			for (int q = 0; q < 8; q++) {
				live[q] = __builtin_fmaf(live[q], p.w, l2[q]);
				vmax[q] = max_f32(vmax[q], l2[q]);
				if (store_row)
					wf_row[64 * (q & 3) + 512 * (q >> 2)] = l2[q] * F_HALF_LOG10_2;
			}
		}

		uint32_t *dst = p.bins + (size_t)((t0 + g0) >> 2) * kN + lane + 256 * w;
#pragma unroll
		for (int q = 0; q < 8; q++)
			dst[64 * (q & 3) + 512 * (q >> 2)] = pack[q];
	}

	float2 *pp = p.partial + (size_t)tile * kN + lane + 256 * w;
#pragma unroll
	for (int q = 0; q < 8; q++)
		pp[64 * (q & 3) + 512 * (q >> 2)] = make_float2(live[q] * F_HALF_LOG10_2,
			(vmax[q] == vmax_init) ? -1000.0f : vmax[q] * F_HALF_LOG10_2);
	}	/* tile loop */
}

/* ------------------------------------------------------------------------ */
/* K1 general N: N/8 threads per spectrum                                    */
/* ------------------------------------------------------------------------ */
/* The reference's plan for any N = 8^k * 2 (fft.cl:397-466 is the N = 1024 instance): k radix-8
 * Stockham passes with p = 1, 8, 64, ... and a final radix-2 pass with p = N/2, N/8 work-items
 * of 8 points each.  One work-group of N/8 threads per spectrum, the N-point exchange slab, the twiddles and
 * the window in dynamic LDS.  Instantiated for N = 1024 with 16-bit bin indices (more than 256 bins): a parity
 * case, not a tuned one (N = 8192 has its own kernel and plan, k1w_fft_bin).
 * Same swizzle phys(e) = e ^ ((e >> 3) & 15): the store patterns of every pass and the
 * lane-contiguous reads stay bank-conflict free for any N (the argument of DESIGN_HISTORY.md only
 * involves address bits 0..6).  Bin indices are 16-bit, 2 spectra per dword. */
static __device__ __forceinline__ int swz(int e) { return e ^ ((e >> 3) & 15); }

template <int LOG2N, bool WRITE_FFT>
__global__ __launch_bounds__((1 << LOG2N) / 8)
void k1big_fft_bin(const K1Params p)
{
	constexpr int N = 1 << LOG2N, T = N / 8, NP8 = LOG2N / 3;
	static_assert(LOG2N % 3 == 1, "plan: radix-8 passes then one radix-2 pass");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	v2f *buf = reinterpret_cast<v2f *>(smem_raw);

	const int i = threadIdx.x;
	const int ntiles = p.total / p.tile;
	/* the whole twiddle table sits behind the exchange slab in LDS: read from global memory it was 21 B per sample of L2 traffic,
	 * against 8 B per sample of IQ.  The reference's layout, 7 per item and pass, then the radix-2 pass's */
	constexpr int TWLEN = ((N / 2 - 8) / 7) * 7 + N / 2;	/* (8 + 64 + ... + N/16) * 7 + N/2 */
	v2f *tws = buf + N;
	float *wins = reinterpret_cast<float *>(tws + TWLEN);	/* and the window behind it: 160 KiB in all at N = 8192 */
	for (int k = i; k < TWLEN; k += T)
		tws[k] = reinterpret_cast<const v2f *>(p.tw)[k];
	for (int k = i; k < N; k += T)
		wins[k] = p.win[k];
	__syncthreads();
	const v2f *twg = tws;
	const v2f s12 = { F_SQRT_1_2, F_SQRT_1_2 };
	const BinConst bk = { p.binA, p.binC, p.amb, p.kappa, p.n_bins, p.thr };
	const float vmax_init = -1000.0f / F_HALF_LOG10_2;
	const float top = (float)(bk.nb - 1);

	for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
	const int t0 = tile * p.tile;
	float live[8], vmax[8];
#pragma unroll
	for (int q = 0; q < 8; q++) { live[q] = 0.0f; vmax[q] = vmax_init; }

	for (int g0 = 0; g0 < p.tile; g0 += 2) {
		uint32_t pack[8];
#pragma unroll
		for (int q = 0; q < 8; q++) pack[q] = 0;

#pragma unroll 1
		for (int u = 0; u < 2; u++) {
			const int t = t0 + g0 + u;
			const float2 *src = p.iq + (size_t)t * p.hop;
			v2f r[8];

			/* window (fft.cl:415-417) */
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const v2f xv = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(src + i + T * j));
				const float wv = wins[i + T * j];
				r[j] = v2f{ xv.x * wv, xv.y * wv };
			}

			/* radix-8 passes p = 1, 8, 64, ... (fft.cl:278-350) */
			int pp = 1;
#pragma unroll
			for (int q8 = 0; q8 < NP8; q8++) {
				const int k = i & (pp - 1);
				if (q8 > 0) {
					const v2f *tw = twg + p.tw_off[q8 - 1] + k * 7;
#pragma unroll
					for (int j = 1; j < 8; j++)
						r[j] = c_mul(r[j], tw[j - 1]);
				}
				dft8(r, s12);
				const int j0 = ((i - k) << 3) + k;
#pragma unroll
				for (int jj = 0; jj < 8; jj++)
					buf[swz(j0 + jj * pp)] = r[R8_PERM(jj)];
				__syncthreads();
				if (q8 + 1 < NP8) {
#pragma unroll
					for (int j = 0; j < 8; j++)
						r[j] = buf[swz(i + T * j)];
					__syncthreads();
				}
				pp <<= 3;
			}

			/* final radix-2 pass, p = N/2 (fft.cl:428-458): butterflies jb = i + T c on (jb, jb + N/2) */
			v2f x[8];
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const int jb = i + T * c;
				v2f a = buf[swz(jb)];
				v2f b = buf[swz(jb + N / 2)];
				b = c_mul(b, twg[p.tw_off[NP8 - 1] + jb]);
				DFT2(a, b);
				x[c] = a;		/* column jb */
				x[c + 4] = b;		/* column jb + N/2 */
			}
			__syncthreads();		/* slab free for the next spectrum */

			if (WRITE_FFT) {
#pragma unroll
				for (int c = 0; c < 4; c++) {
					reinterpret_cast<v2f *>(p.fft_out)[(size_t)t * N + i + T * c] = x[c];
					reinterpret_cast<v2f *>(p.fft_out)[(size_t)t * N + i + T * c + N / 2] = x[c + 4];
				}
			}

			/* epilogue (display.cl:136,161-168), as in the 1024-point kernels, 16-bit bin indices */
			float l2[8];
			uint32_t bn[8];
			uint32_t amb = 0;
#pragma unroll
			for (int q = 0; q < 8; q++) {
				uint32_t ab;
				const float rr = bin_fast(x[q].x, x[q].y, bk, &l2[q], &ab);
				amb = amb > ab ? amb : ab;
				bn[q] = (uint32_t)(int)__builtin_amdgcn_fmed3f(rr, 0.0f, top);
			}
			if (amb > __float_as_uint(bk.amb)) {
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const float v = __builtin_fmaf(bk.A, l2[q], bk.C);
					const float rr = __builtin_rintf(v);
					const float a = __builtin_fmaf(__builtin_fabsf(l2[q]), bk.kappa, __builtin_fabsf(v - rr));
					if (!(a <= bk.amb)) {
						float nl2;
						bn[q] = bin_exact(x[q].x, x[q].y, l2[q], (int)bn[q], ThrScalar{ bk.thr }, bk.nb, &nl2);
						l2[q] = nl2;
					}
				}
			}
			const bool store_row = (t >= p.wf_first);
			float *wf_row = p.wf + (size_t)((p.wf_pos0 + t) & p.wf_mask) * N + i;
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const int col_off = T * (q & 3) + (N / 2) * (q >> 2);
				pack[q] |= bn[q] << (16 * u);
				live[q] = __builtin_fmaf(live[q], p.w, l2[q]);
				vmax[q] = max_f32(vmax[q], l2[q]);
				if (store_row)
					wf_row[col_off] = l2[q] * F_HALF_LOG10_2;
			}
		}
		uint32_t *dst = p.bins + (size_t)((t0 + g0) >> 1) * N + i;
#pragma unroll
		for (int q = 0; q < 8; q++)
			dst[T * (q & 3) + (N / 2) * (q >> 2)] = pack[q];
	}
	float2 *pp2 = p.partial + (size_t)tile * N + i;
#pragma unroll
	for (int q = 0; q < 8; q++)
		pp2[T * (q & 3) + (N / 2) * (q >> 2)] = make_float2(live[q] * F_HALF_LOG10_2,
			(vmax[q] == vmax_init) ? -1000.0f : vmax[q] * F_HALF_LOG10_2);
	}
}

/* Buffer addressing for the 8192- and 65536-point kernels: every global access of their loops is `scalar base (descriptor) + ONE 32-bit per-lane
 * offset + a scalar offset` -- buffer_load / buffer_store ... offen -- where the per-lane offset is fixed for the kernel's lifetime and
 * everything that changes (spectrum, row, column block c) is scalar arithmetic.  With plain pointers the compiler folded the
 * column-block constants into 64-bit per-lane adds (240 of them per spectrum) and spilled.  Arrays addressed this way are < 4 GiB. */
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base)
{
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffffffff, 0x00020000);	/* raw buffer, 32-bit data format */
}
constexpr int kAuxNT = 2, kAuxSC1 = 16;		/* gfx94x / gfx950 cache-policy bits of the buffer intrinsics: nt, sc1 */
#ifndef K1H_OUT_AUX
#define K1H_OUT_AUX 2				/* cache policy of the 65536-point kernel's row / index stores (A/B builds) */
#endif
#ifndef K1H_IQ_MOD				/* ... and of its LDS-DMA of the IQ (K1H_IQ_POL: A/B builds) */
#if !defined(K1H_IQ_POL) || K1H_IQ_POL == 0
#define K1H_IQ_MOD "nt"
#elif K1H_IQ_POL == 1
#define K1H_IQ_MOD ""
#elif K1H_IQ_POL == 2
#define K1H_IQ_MOD "sc1"
#elif K1H_IQ_POL == 3
#define K1H_IQ_MOD "sc0 sc1"
#elif K1H_IQ_POL == 4
#define K1H_IQ_MOD "sc0 sc1 nt"
#elif K1H_IQ_POL == 5
#define K1H_IQ_MOD "sc1 nt"
#else
#define K1H_IQ_MOD "sc0 nt"
#endif
#endif
template <int AUX>
static __device__ __forceinline__ void bst_v2f(v2f v, __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
	__builtin_amdgcn_raw_buffer_store_b64(u2v{ __float_as_uint(v.x), __float_as_uint(v.y) }, rs, voff, soff, AUX);
}
template <int AUX>
static __device__ __forceinline__ v2f bld_v2f(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff)
{
	const u2v u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX);
	return v2f{ __uint_as_float(u.x), __uint_as_float(u.y) };
}

/* X[jj] of a radix-16 pass sits in r[bitrev4(jj)] */
#define R16_PERM(jj) ((((jj) & 1) << 3) | (((jj) & 2) << 1) | (((jj) & 4) >> 1) | (((jj) & 8) >> 3))

/* ------------------------------------------------------------------------ */
/* K1 for N = 8192: 512 threads per spectrum, 16 points per thread            */
/* ------------------------------------------------------------------------ */
/* The oracle's plan at this length (oracle/fosphor_oracle.c: o_pass_radix16_fma x 3 + o_pass_radix2_fma; no reference behaviour
 * exists beyond N = 1024): Stockham radix-16 passes p = 1, 16, 256 over 512 work-items of 16 points -- ONE item per thread in every
 * pass -- and the radix-2 pass p = 4096 of fft.cl:428-458.  Round 5 measured that this kernel's LDS is as busy as its VALUs (three
 * exchanges of 64 KiB each way per spectrum = 2.2 us per CU next to 2.0 us of VALU issue, and the two add up: profiles/r05_lds.md);
 * against the radix 8.8.8.8.2 form it replaced this plan moves TWO AND A HALF exchanges through the LDS:
 *   - exchanges 1 and 2 (behind the passes p = 1 and p = 16) are full: 16 stores, one barrier, 16 loads per thread (two 64 KiB slabs
 *     used alternately: the barrier behind the stores of an exchange also proves that every thread has finished the loads of the
 *     exchange before the previous one, i.e. of the slab written next);
 *   - behind the pass p = 256 item i = k + 256 h holds X3[4096 h + k + 256 m], m < 16, and the radix-2 butterflies pair (jb, jb + 4096):
 *     thread (h, k) keeps its eight outputs m in [8 h, 8 h + 8), hands the other eight to thread (1 - h, k) through the LDS and does the
 *     butterflies jb = k + 256 m of its half: 8 stores + 8 loads per thread.  Its columns are k + 2048 h + 256 c + 4096 v, c < 8, v < 2;
 *   - LDS swizzle phys(e) = e ^ ((e >> 4) & 31) (8-byte elements): the three store patterns (16 i + m; 256 (i >> 4) + (i & 15) + 16 m;
 *     4096 h + k + 256 m) put the 16 lanes of a ds_write_b64 group into 16 different bank pairs, the lane-contiguous loads e = i + 512 j
 *     the 32 lanes of a ds_read_b64 group into 32; each store is `per-thread constant ^ compile-time constant`, each load an immediate;
 *   - everything a thread needs from the tables is fixed per thread and sits in registers: 16 window taps (as the 8 pairs of the first
 *     pass's stage-A butterflies), 2 x 8 twiddles of the passes p = 16, 256, the 8 of its radix-2 butterflies;
 *   - the overlap of overlap_cc (overlap_cc_impl.cc:64-79) lives in REGISTERS: a thread holds the raw IQ of elements th + 512 j, j < 16;
 *     the next window of a tile starts hop = N / R samples later, i.e. 16 / R rows of 512 -- its row j is this window's row j + 16 / R
 *     of the same thread.  With R = 2 (BASELINE C3) a spectrum costs eight 8-byte loads per thread: every sample of the stream is fetched
 *     once.  Any hop works (8-byte loads need no 16-byte alignment): odd hops reload all sixteen rows.
 * One work-group (8 waves, <= 256 registers) per CU.  16-bit bin indices, 2 spectra per dword. */
/* Work-group barrier for exchanges through LDS only: waits for this wave's LDS operations, not for its outstanding
 * global loads and stores (__syncthreads() also drains vmcnt, which would park every wave of the work-group behind the
 * IQ requested for the NEXT spectrum).  Nothing is handed from thread to thread through global memory in these kernels. */
static __device__ __forceinline__ void wg_barrier_lds()
{
#if defined(FOSPHOR_AMD_PROBES) && defined(K1W_NOSYNC)
	asm volatile("" ::: "memory");		/* timing probe only: results are garbage */
#else
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

/* K1W_TIMING=1 (probe builds only, tools/k1w_phase_timing.py): s_memtime stamps per phase of the 8192-point kernel's loop, accumulated per
 * wave (waves 0 and 4 of a work-group: the early and the late one of a SIMD) into K1Params::dbg[(work-group * 2 + slot) * 16 + phase].
 * Reading the clock waits for the wave's LDS operations (s_memtime answers on lgkmcnt). */
#ifndef K1W_TIMING
#define K1W_TIMING 0
#endif
/* K1W_PROBE (probe builds, results are garbage: timing only): 1 no LDS stores, 2 no LDS loads, 4 no epilogue, 8 no index stores,
 * 16 no IQ requests inside the loop */
#if defined(FOSPHOR_AMD_PROBES) && defined(K1W_PROBE)
#define K1W_P(b) ((K1W_PROBE) & (b))
#else
#define K1W_P(b) 0
#endif
#ifndef K1W_READ_FIRST
#define K1W_READ_FIRST 0		/* (A/B builds) 1: a late wave requests its operands BEFORE its epilogue piece (measured: 1.2 % slower) */
#endif
#ifndef K1W_PRIO
#define K1W_PRIO 1			/* (A/B builds) 1: the passes run at a higher issue priority than the epilogue pieces */
#endif
#if K1W_TIMING
#define K1W_STAMP(i) do { const uint32_t _now = (uint32_t)__builtin_readcyclecounter(); wacc[i] += _now - wprev; wprev = _now; } while (0)
#else
#define K1W_STAMP(i) do { } while (0)
#endif

constexpr int kK1wIdxStores = 16;	/* index stores a thread issues per ODD spectrum (one dword per column and pair of spectra): the immediate of the
					 * hand-written wait for the IQ requested before them */
template <int SHIFT>
__global__ __launch_bounds__(512, 2)
void k1w_fft_bin(const K1Params p)
{
	constexpr int N = 8192, TH = 512;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	/* two 64 KiB slabs; a spectrum's exchanges use A, B, A (the half one: its first 32 KiB) and the next spectrum's B, A, B */
	v2f *slab0 = reinterpret_cast<v2f *>(smem_raw);
	v2f *slab1 = slab0 + N;
	/* behind the slabs: the exact-bin thresholds (n_bins + 1 <= 513 doubles): the rare path that consults them must not wait behind the IQ
	 * in flight (a table load through the vector memory path returns in order), and while one wave is in it the other seven wait at the
	 * next barrier */
	typedef const __attribute__((address_space(3))) double *lds_cdp;
	double *thr_g = reinterpret_cast<double *>(slab0 + 2 * N);
	const lds_cdp thr_l = (lds_cdp)thr_g;

	const int th = threadIdx.x;
	const int ntiles = p.total / p.tile;
	const v2f *twg = reinterpret_cast<const v2f *>(p.tw);
	const v2f two = { 2.0f, 2.0f };
	const BinConst bk = { p.binA, p.binC, p.amb, p.kappa, p.n_bins, p.thr };
	const float vmax_init = -1000.0f / F_HALF_LOG10_2;
	const float top = (float)(bk.nb - 1);

	/* ---- per-thread constants ------------------------------------------------ */
	const int hh = th >> 8, kk = th & 255;		/* item th = kk + 256 hh of the pass p = 256 */
	const int hu = __builtin_amdgcn_readfirstlane(th >> 8);	/* = hh, as a scalar (wave-uniform: waves 0-3 / 4-7), for the whole kernel: taken inside the spectrum
								 * loop it kept th >> 8 alive in a vector register -- the one the general-hop form spilled */
	const v2f w16c = twg[p.tw_off[0]], w8c = twg[p.tw_off[0] + 1], w163c = twg[p.tw_off[0] + 2];	/* W16, W8, W16^3: the first pass */
	v2f wab[8];			/* taps of elements th + 512 j and th + 512 (j + 8): the pair of a first-pass stage-A butterfly */
	v2f tw16[8], tw256[8];		/* w^8, w^4, w^2, w^2 W8, w, w W16, w W8, w W16^3 for k = th & 15, th & 255 */
	v2f twr[8];			/* radix-2 twiddles k = kk + 256 (8 hh + c) */
#pragma unroll
	for (int j = 0; j < 8; j++) {
		wab[j] = v2f{ p.win[th + 512 * j], p.win[th + 512 * (j + 8)] };
		twr[j] = twg[p.tw_off[3] + kk + 256 * (8 * hh + j)];
	}
	for (int e = th; e <= p.n_bins && e < 520; e += TH)
		thr_g[e] = p.thr[e];
	__syncthreads();
#pragma unroll
	for (int n = 0; n < 8; n++) {
		tw16[n]  = twg[p.tw_off[1] + (th & 15) * 8 + n];
		tw256[n] = twg[p.tw_off[2] + kk * 8 + n];
	}

	/* ---- LDS addressing (8-byte elements, phys(e) = e ^ ((e >> 4) & 31)) ----
	 * loads of every pass: e = th + 512 j -> phys = rd + 512 j
	 * stores: pass p = 1    e = 16 th + m                          -> st1 ^ m
	 *         pass p = 16   e = 256 (th >> 4) + (th & 15) + 16 m   -> st2 ^ ((m ^ 16 (m & 1)) | 32 (m >> 1))
	 *         half exchange (plain layout [m''][th]: lane-contiguous both ways)  stores m'' 512 + th, loads m'' 512 + (th ^ 256) */
	const int rd  = th ^ ((th >> 4) & 31);
	const int st1 = (32 * (th >> 1)) | ((16 * (th & 1)) ^ (th & 31));
	const int st2 = (256 * (th >> 4)) | ((th & 15) ^ (16 * ((th >> 4) & 1)));

	/* SHIFT = 16 / R for hop = N / R, R = 2, 4, 8, 16: the next window's row j is this window's row j + SHIFT of the same thread;
	 * SHIFT = 16: any other hop, every row is requested again */
	const uint32_t iq_vo = 8u * (uint32_t)th;		/* element th + 512 j of a window at 8 th + 4096 j (scalar descriptor + one lane offset) */
	auto ld_iq = [&](__amdgpu_buffer_rsrc_t rs, int j) __attribute__((always_inline)) -> v2f {
		return bld_v2f<kAuxNT>(rs, iq_vo, 4096u * (uint32_t)j);
	};

	v2f q[16];			/* raw IQ of the spectrum to be processed next: rows th + 512 j */
	/* column of xo[m]: cb + 256 (m & 7) + 4096 (m >> 3), cb = kk + 2048 hh.  ONE register carries it through the spectrum loop, as the
	 * byte offset 2 cb of the column's short in an index row (the kernel has no register to spare: tools/check_k1w_loads.py); the rare
	 * users of cb itself (waterfall rows, the bytes of 9th bits, the tile's partials) take it back out of it where they run */
	const uint32_t cb2 = 2u * ((uint32_t)kk + 2048u * (uint32_t)hh);
#define K1W_CB() ({ uint32_t _c; asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(_c) : "v"(cb2)); _c; })

	/* Epilogue of columns [M0, M1) of spectrum tp, whose FFT is in xo: log-power, exact 16-bit bin, live / max, waterfall row
	 * (display.cl:136-150,161-168).  Per column, nothing carried from column to column: it is cut into three pieces that
	 * run between the LDS stores of the NEXT spectrum's exchanges and the barrier behind them, i.e. while this wave
	 * would otherwise wait for the slowest one. */
#ifndef K1W_P1
#define K1W_P1 6		/* the three epilogue pieces: columns [0, P1), [P1, P2), [P2, 16) of a thread (A/B builds) */
#define K1W_P2 11
#endif
#define K1W_COL(m) (256 * ((m) & 7) + 4096 * ((m) >> 3))
#define K1W_EPI(M0, M1, tp) do { \
		if (K1W_P(4)) break; \
		if (K1W_PRIO) __builtin_amdgcn_s_setprio(0); \
		const bool _row = ((tp) >= p.wf_first); \
		float *_wfr = p.wf + (size_t)((p.wf_pos0 + (tp)) & p.wf_mask) * N; \
		/* index stores (512 bins: 9 bits), 1.125 B per sample instead of the 2 B of a 16-bit index (round 6): \
		 *   low bytes   one SHORT per column and PAIR of spectra, [t / 2][column] (even spectrum in the low byte) \
		 *   9th bits    one BYTE per column and EIGHT spectra, [t / 8][column] behind the shorts (bit u = spectrum 8 (t / 8) + u) \
		 * A vector-memory instruction costs a CU 8-17 cycles whatever it carries (tools/ubench/vmem_rate.hip: 8.2 for 64 dense shorts, \
		 * 10.8 for 64 dwords), and with a store per sample the index stores were a quarter of this kernel's time: the low bytes of an \
		 * even spectrum wait in four registers (four columns each) for the odd one's, the 9th bits of eight spectra in four more \
		 * (tiles are multiples of 8: launch_k1).  Scalar base (SALU) + ONE lane offset + immediate; column cb + K1W_COL(m) of a row: \
		 * shorts at 2 cb + 512 (m & 7) + 8192 (m >> 3), bytes at cb + 256 (m & 7) + 4096 (m >> 3) */ \
		const char *_blo = reinterpret_cast<const char *>(p.bins) + (size_t)((tp) >> 1) * (N * 2); \
		const char *_bhi = reinterpret_cast<const char *>(p.bins) + (size_t)p.total * N + (size_t)((tp) >> 3) * N; \
		const uint32_t _bo2 = cb2; \
		const uint32_t _sh = (uint32_t)(tp) & 7u;		/* uniform */ \
		if ((M0) == 0 && _sh == 0) { hi9[0] = 0; hi9[1] = 0; hi9[2] = 0; hi9[3] = 0; } \
		float _l2[(M1) - (M0)]; uint32_t _bn[(M1) - (M0)]; uint32_t _amb = 0; \
		_Pragma("unroll") \
		for (int m = (M0); m < (M1); m++) { \
			uint32_t ab; \
			const float rr = bin_fast(xo[m].x, xo[m].y, bk, &_l2[m - (M0)], &ab); \
			_amb = _amb > ab ? _amb : ab;		/* v_max_u32: NaN / inf order above every finite measure */ \
			_bn[m - (M0)] = (uint32_t)(int)__builtin_amdgcn_fmed3f(rr, 0.0f, top); \
		} \
		/* ONE branch per piece (a compare + exec save + branch per sample cost 9 % of this kernel): rare -- find the samples again \
		 * and decide them against the exact thresholds */ \
		if (!K1_DBG_NO_EXACT && _amb > __float_as_uint(bk.amb)) { \
			_Pragma("unroll") \
			for (int m = (M0); m < (M1); m++) { \
				const float v = __builtin_fmaf(bk.A, _l2[m - (M0)], bk.C); \
				const float a = __builtin_fmaf(__builtin_fabsf(_l2[m - (M0)]), bk.kappa, __builtin_fabsf(v - __builtin_rintf(v))); \
				if (!(a <= bk.amb)) { \
					float nl2; \
					_bn[m - (M0)] = bin_exact(xo[m].x, xo[m].y, _l2[m - (M0)], (int)_bn[m - (M0)], thr_l, bk.nb, &nl2); \
					_l2[m - (M0)] = nl2; \
				} \
			} \
		} \
		_Pragma("unroll") \
		for (int m = (M0); m < (M1); m++)		/* the 9th bit joins its column's byte: bit (t & 7) */ \
			hi9[m >> 2] = (__builtin_amdgcn_ubfe(_bn[m - (M0)], 8, 1) << (8 * (m & 3) + _sh)) | hi9[m >> 2]; \
		if (!((tp) & 1)) {		/* (uniform: ONE branch per piece) even spectrum: keep the low bytes, four columns per register */ \
			_Pragma("unroll") \
			for (int m = (M0); m < (M1); m++) \
				held[m >> 2] = __builtin_amdgcn_perm(_bn[m - (M0)], held[m >> 2], \
				                                     (m & 3) == 0 ? 0x03020104u : (m & 3) == 1 ? 0x03020400u : (m & 3) == 2 ? 0x03040100u : 0x04020100u); \
		} else if (!K1W_P(8)) {		/* odd spectrum: the short of both */ \
			_Pragma("unroll") \
			for (int m = (M0); m < (M1); m++) { \
				const char *_sb = _blo + 8192 * (m >> 3); \
				const uint32_t _d = __builtin_amdgcn_perm(_bn[m - (M0)], held[m >> 2], 0x0c0c0400u | (uint32_t)(m & 3)); \
				switch (m & 7) { \
				case 0:  asm volatile("global_store_short %0, %1, %2" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				case 1:  asm volatile("global_store_short %0, %1, %2 offset:512" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				case 2:  asm volatile("global_store_short %0, %1, %2 offset:1024" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				case 3:  asm volatile("global_store_short %0, %1, %2 offset:1536" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				case 4:  asm volatile("global_store_short %0, %1, %2 offset:2048" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				case 5:  asm volatile("global_store_short %0, %1, %2 offset:2560" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				case 6:  asm volatile("global_store_short %0, %1, %2 offset:3072" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				default: asm volatile("global_store_short %0, %1, %2 offset:3584" :: "v"(_bo2), "v"(_d), "s"(_sb) : "memory"); break; \
				} \
			} \
			if (_sh == 7) {		/* (uniform) the eighth spectrum: the bytes of 9th bits, scalar base + one lane offset + immediate like the shorts; \
						 * byte 0 / 2 of a register as it is (global_store_byte / _d16_hi), byte 1 / 3 of its copy shifted by 8 */ \
				const uint32_t cb = K1W_CB(); \
				_Pragma("unroll") \
				for (int m = (M0); m < (M1); m++) { \
					const char *_hb = _bhi + 4096 * (m >> 3); \
					const uint32_t _hv = (m & 1) ? (hi9[m >> 2] >> 8) : hi9[m >> 2]; \
					if (m & 2) { \
						switch (m & 7) { \
						case 2:  asm volatile("global_store_byte_d16_hi %0, %1, %2 offset:512" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						case 3:  asm volatile("global_store_byte_d16_hi %0, %1, %2 offset:768" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						case 6:  asm volatile("global_store_byte_d16_hi %0, %1, %2 offset:1536" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						default: asm volatile("global_store_byte_d16_hi %0, %1, %2 offset:1792" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						} \
					} else { \
						switch (m & 7) { \
						case 0:  asm volatile("global_store_byte %0, %1, %2" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						case 1:  asm volatile("global_store_byte %0, %1, %2 offset:256" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						case 4:  asm volatile("global_store_byte %0, %1, %2 offset:1024" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						default: asm volatile("global_store_byte %0, %1, %2 offset:1280" :: "v"(cb), "v"(_hv), "s"(_hb) : "memory"); break; \
						} \
					} \
				} \
			} \
		} \
		_Pragma("unroll") \
		for (int m = (M0); m < (M1); m++) { \
			live[m] = __builtin_fmaf(live[m], p.w, _l2[m - (M0)]); \
			vmax[m] = max_f32(vmax[m], _l2[m - (M0)]); \
		} \
		if (_row) {		/* uniform, rare (the last wf_rows spectra of a call): one branch per piece instead of one per sample; the row \
					 * values are recomputed from the log-powers, which the live / max updates above kept alive anyway */ \
			float *_wf = _wfr + K1W_CB(); \
			_Pragma("unroll") \
			for (int m = (M0); m < (M1); m++) \
				_wf[K1W_COL(m)] = _l2[m - (M0)] * F_HALF_LOG10_2; \
		} \
		if (K1W_PRIO) __builtin_amdgcn_s_setprio(2); \
	} while (0)

	if (K1W_PRIO) __builtin_amdgcn_s_setprio(2);
#if K1W_TIMING
	uint32_t wacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t wprev = (uint32_t)__builtin_readcyclecounter();
#endif
	for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
	const int t0 = tile * p.tile;
	float live[16], vmax[16];
	uint32_t held[4] = { 0, 0, 0, 0 };		/* low bytes of the tile's last even spectrum's bin indices, four columns per register */
	uint32_t hi9[4] = { 0, 0, 0, 0 };		/* 9th bits of the indices of up to eight spectra, one byte per column, four columns per register */
#pragma unroll
	for (int m = 0; m < 16; m++) { live[m] = 0.0f; vmax[m] = vmax_init; }

	{
		const __amdgpu_buffer_rsrc_t src = make_rsrc(p.iq + (size_t)t0 * p.hop);
#pragma unroll
		for (int j = 0; j < 16; j++)
			q[j] = ld_iq(src, j);
	}

	v2f xo[16];			/* FFT of the previous spectrum of the tile, its epilogue still to do */
#pragma unroll
	for (int m = 0; m < 16; m++) xo[m] = v2f{ 0.0f, 0.0f };

#pragma unroll 1
	for (int g = 0; g < p.tile; g++) {
		const int t = t0 + g;
		const bool have_prev = g > 0;			/* uniform */
		/* The two waves of a SIMD (waves w and w + 4 of the work-group) run their epilogue pieces on opposite sides of the barrier:
		 * one computes while the other waits for its LDS loads, instead of all eight moving from LDS to VALU and back together */
		const bool late = hu != 0;
		v2f x[16];
		{ v2f *sw = slab0; slab0 = slab1; slab1 = sw; }		/* (the first spectrum starts on the second slab) */

		/* x[j] = element th + 512 j (the window multiply of fft.cl:415-417 rides on the first pass) */
		/* ---- pass 1: p = 1, item th, outputs e = 16 th + m -> slab0.  Before the next spectrum's IQ is requested: the requests then
		 * land in the registers this pass has just consumed (requested first, they needed sixteen more and a copy at the end of the loop) ---- */
		/* the IQ requested one iteration ago has arrived once at most the index stores issued BEHIND the requests are outstanding: the
		 * sixteen of an odd spectrum's epilogue, which ran in the previous iteration if that one's g was even and >= 2 (more, if waterfall
		 * rows or fft_out went out as well: the wait is then longer than needed, not shorter) */
#define K1W_Q16 "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), \
		"+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15])
		/* (ONE statement, the choice inside it: two statements under an if made the compiler copy q -- before the wait) */
		/* (kK1wIdxStores: ONE constant for the wait's immediate and for what K1W_EPI issues per odd spectrum -- a change of the index
		 * format that packs the stores must change both; tools/check_k1w_loads.py counts the stores of the compiled loop against it) */
		static_assert(kK1wIdxStores == 16, "the counted wait below and K1W_EPI's index stores (one dword per column and pair of spectra) go together");
		asm volatile("s_cmp_eq_u32 %16, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%17)\n\ts_branch 2f\n1:\ts_waitcnt vmcnt(0)\n2:"
		             : K1W_Q16 : "s"(__builtin_amdgcn_readfirstlane((!K1W_P(8) && (g & 1) && g >= 3) ? 1 : 0)), "n"(kK1wIdxStores) : "scc");
#undef K1W_Q16
#pragma unroll
		for (int j = 0; j < 16; j++)
			x[j] = q[j];
		K1W_STAMP(0);			/* radix 2 of the previous spectrum, loop overhead, wait for the IQ */
		pass16_first<true>(x, wab, w16c, w8c, w163c, two);
		K1W_STAMP(1);

		/* raw IQ of the next spectrum of this tile: shared rows move down, the new ones are requested now.  UNCONDITIONALLY (behind the
		 * tile's last spectrum: of that spectrum again, unused): a load inside a branch whose result merges with an older value at the
		 * join makes the compiler wait for it right there */
		{
			const int tn = (g + 1 < p.tile) ? t + 1 : t;
			const __amdgpu_buffer_rsrc_t src = make_rsrc(p.iq + (size_t)tn * p.hop);
			/* (moves the compiler cannot sink: left to it, they went behind the requests -- whose results then needed registers of their
			 * own, a copy at the end of the loop and, for that copy, a wait for every store issued in between) */
#pragma unroll
			for (int j = 0; j < 16 - SHIFT; j++)
				asm volatile("v_mov_b64 %0, %1" : "=v"(q[j]) : "v"(q[j + SHIFT]));
			/* The requests are made by hand, and so is the wait for them at the top of the next iteration: loads and stores leave the
			 * vmcnt queue IN ORDER, and the wait the compiler places for loads it knows about -- vmcnt(0) -- also sat through the
			 * acknowledgement of every index store issued since (a third of this kernel's time: probe builds without the stores / without
			 * the requests, profiles/r05_c3.md) */
#pragma unroll
			for (int j = 16 - SHIFT; j < 16; j++)
				if (!K1W_P(16))
					asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen nt" : "=v"(q[j]) : "v"(iq_vo), "s"(src), "s"(4096u * (uint32_t)j));
		}

#pragma unroll
		for (int m = 0; m < 16; m++)
			if (!K1W_P(1)) slab0[st1 ^ m] = x[R16_PERM(m)];
		if (have_prev && !late) K1W_EPI(0, K1W_P1, t - 1);
		K1W_STAMP(2);			/* IQ requests, stores (until done), early piece */
		wg_barrier_lds();
		K1W_STAMP(3);			/* barrier */
#if K1W_READ_FIRST
		/* the reads are requested BEFORE the late piece: a late wave's piece then runs while its operands travel (and while the early
		 * wave of its SIMD, whose reads were requested at the same moment, has nothing to compute yet) */
#pragma unroll
		for (int j = 0; j < 16; j++)
			if (!K1W_P(2)) x[j] = slab0[rd + 512 * j];
		if (have_prev && late) K1W_EPI(0, K1W_P1, t - 1);
		K1W_STAMP(4);			/* late piece */
#else
		if (have_prev && late) K1W_EPI(0, K1W_P1, t - 1);
		K1W_STAMP(4);			/* late piece */
#pragma unroll
		for (int j = 0; j < 16; j++)
			if (!K1W_P(2)) x[j] = slab0[rd + 512 * j];
#endif
		K1W_STAMP(5);			/* reads (until all have arrived) */

		/* ---- pass 2: p = 16, k = th & 15, outputs e = 256 (th >> 4) + (th & 15) + 16 m -> slab1 ---- */
		pass16_ab<true>(x, tw16[0], tw16[1], two);
		pass16_cd<true>(x, tw16[2], tw16[3], tw16[4], tw16[5], tw16[6], tw16[7], two);
		K1W_STAMP(6);			/* pass 2 */
#pragma unroll
		for (int m = 0; m < 16; m++)
			if (!K1W_P(1)) slab1[st2 ^ ((m ^ (16 * (m & 1))) | (32 * (m >> 1)))] = x[R16_PERM(m)];
		if (have_prev && !late) K1W_EPI(K1W_P1, K1W_P2, t - 1);
		K1W_STAMP(7);
		wg_barrier_lds();
		K1W_STAMP(8);
#if K1W_READ_FIRST
#pragma unroll
		for (int j = 0; j < 16; j++)
			if (!K1W_P(2)) x[j] = slab1[rd + 512 * j];
		if (have_prev && late) K1W_EPI(K1W_P1, K1W_P2, t - 1);
		K1W_STAMP(9);
#else
		if (have_prev && late) K1W_EPI(K1W_P1, K1W_P2, t - 1);
		K1W_STAMP(9);
#pragma unroll
		for (int j = 0; j < 16; j++)
			if (!K1W_P(2)) x[j] = slab1[rd + 512 * j];
#endif
		K1W_STAMP(10);

		/* ---- pass 3: p = 256, k = kk: X3[4096 hh + kk + 256 m] = x[R16_PERM(m)]; the half this thread's butterflies do not need goes to
		 * thread th ^ 256 through slab0 ([m''][th]: m'' = m - 8 (1 - hh)) ---- */
		pass16_ab<true>(x, tw256[0], tw256[1], two);
		pass16_cd<true>(x, tw256[2], tw256[3], tw256[4], tw256[5], tw256[6], tw256[7], two);
		K1W_STAMP(11);			/* pass 3 */
		if (hu == 0) {			/* uniform per wave (waves 0-3 / 4-7): a scalar branch */
#pragma unroll
			for (int c = 0; c < 8; c++)
				if (!K1W_P(1)) slab0[512 * c + th] = x[R16_PERM(8 + c)];
		} else {
#pragma unroll
			for (int c = 0; c < 8; c++)
				if (!K1W_P(1)) slab0[512 * c + th] = x[R16_PERM(c)];
		}
		if (have_prev && !late) K1W_EPI(K1W_P2, 16, t - 1);
		K1W_STAMP(12);
		wg_barrier_lds();
		K1W_STAMP(13);
		/* ---- radix 2, p = 4096 (fft.cl:428-458; o_pass_radix2_fma): (jb, jb + 4096), jb = kk + 256 (8 hh + c) ->
		 * xo[c] = X[jb], xo[c + 8] = X[jb + 4096] ---- */
		{
			v2f o[8];
#if K1W_READ_FIRST
#pragma unroll
			for (int c = 0; c < 8; c++)
				o[c] = K1W_P(2) ? x[c] : slab0[512 * c + (th ^ 256)];
			if (have_prev && late) K1W_EPI(K1W_P2, 16, t - 1);
			K1W_STAMP(14);
#else
			if (have_prev && late) K1W_EPI(K1W_P2, 16, t - 1);
			K1W_STAMP(14);
#pragma unroll
			for (int c = 0; c < 8; c++)
				o[c] = K1W_P(2) ? x[c] : slab0[512 * c + (th ^ 256)];
#endif
			K1W_STAMP(15);
			/* (step by step, like bf8; the two forms differ in where a and b come from) */
#define K1W_R2(A, B) do { \
				v2f u[8], pa[8]; \
				_Pragma("unroll") \
				for (int c = 0; c < 8; c++) \
					asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(u[c]) : "v"(B), "v"(twr[c]), "v"(A)); \
				_Pragma("unroll") \
				for (int c = 0; c < 8; c++) \
					asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(pa[c]) : "v"(B), "v"(twr[c]), "v"(u[c])); \
				_Pragma("unroll") \
				for (int c = 0; c < 8; c++) \
					asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(xo[c + 8]) : "v"(A), "s"(two), "v"(pa[c])); \
				_Pragma("unroll") \
				for (int c = 0; c < 8; c++) \
					xo[c] = pa[c]; \
			} while (0)
			if (hu == 0)			/* X3[jb] is this item's output m = c, X3[jb + 4096] item th + 256's */
				K1W_R2(x[R16_PERM(c)], o[c]);
			else				/* X3[jb] is item th - 256's output m = 8 + c, X3[jb + 4096] this item's */
				K1W_R2(o[c], x[R16_PERM(8 + c)]);
#undef K1W_R2
		}

		if (p.fft_out) {		/* (tests) */
#pragma unroll
			for (int m = 0; m < 16; m++)
				reinterpret_cast<v2f *>(p.fft_out)[(size_t)t * N + K1W_CB() + K1W_COL(m)] = xo[m];
		}
	}
	/* the last iteration's requests (made unconditionally, see above) still own their registers: nothing may reuse them before they have landed */
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]),
	             "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]));
	K1W_EPI(0, 16, t0 + p.tile - 1);		/* the tile's last spectrum */

	float2 *pp2 = p.partial + (size_t)tile * N + K1W_CB();
#pragma unroll
	for (int m = 0; m < 16; m++)
		pp2[K1W_COL(m)] = make_float2(live[m] * F_HALF_LOG10_2,
			(vmax[m] == vmax_init) ? -1000.0f : vmax[m] * F_HALF_LOG10_2);
	}
#if K1W_TIMING
	if (p.dbg && (th & 255) == 0) {
#pragma unroll
		for (int i = 0; i < 16; i++)
			p.dbg[((size_t)blockIdx.x * 2 + (th >> 8)) * 16 + i] = wacc[i];
	}
#endif
#undef K1W_EPI
#undef K1W_CB
#undef K1W_COL
}

/* ------------------------------------------------------------------------ */
/* K1 for N = 65536: radix-16 plan, two stages, the intermediate in the XCD's L2 */
/* ------------------------------------------------------------------------ */
/* No reference behaviour exists at N = 65536 (fft.cl has one length, 1024): the plan is this build's own and the oracle
 * restates it (oracle/fosphor_oracle.c, o_pass_radix16_fma): four Stockham radix-16 passes, p = 1, 16, 256, 4096,
 * 4096 virtual work-items of 16 points, a pass = four radix-2 stages with the twiddles on the butterflies (bf(), above).  512 KiB per
 * spectrum does not fit one CU's LDS, but the data flow factors into two 256-point levels:
 *
 *   stage A  passes 1-2 only mix inputs whose index is congruent mod 256: for each residue q they ARE the two passes of a
 *            256-point transform on x[q + 256 m], m < 256, and leave its 256 results at w[256 q + kk], kk < 256.
 *            SIXTEEN LANES do one such transform (16 points each, one 16 x 16 transpose through LDS in between), so a
 *            wavefront does four residues ON ITS OWN: no work-group barrier anywhere in stage A.
 *   stage B  passes 3-4 only mix elements with the same offset kk: for each kk a 256-point transform over w[256 q + kk],
 *            q < 256, whose outputs are columns kk + 256 jj3 + 4096 jj4.  A work-group takes 32 adjacent offsets with
 *            thread = (offset, item): every global access of the stage -- the intermediate coming in, rows, bin indices
 *            and partials going out -- is a run of 32 consecutive columns (128 / 256 B), and the one exchange between its
 *            two passes goes through a work-group-wide LDS array behind ONE barrier.
 *
 * A CLUSTER of 8 work-groups on ONE XCD takes a spectrum through both stages (member m: residues [32 m, 32 m + 32) in stage A,
 * offsets [32 m, 32 m + 32) in stage B); between the stages the spectrum makes one round trip through the XCD's L2 (plain
 * stores + s_waitcnt vmcnt(0) + relaxed agent-scope atomics: the L2 is the coherence point of its CUs), laid out
 * [offset / 32][residue][offset % 32] so that both sides move whole 128-byte runs.  Clusters form from XCC_ID tickets and
 * claim tiles dynamically (progress never depends on a work-group that is not resident); every wait on another work-group
 * is bounded and ends in an error word the host turns into -EIO.
 *
 * Against the radix-8 form this replaces (8.8.8 | 8.8.2, 1024 threads of 8 points, eight work-group barriers per spectrum):
 * 512 threads of 16 points, ONE work-group barrier per spectrum besides the cluster hand-off, twiddles of a thread fixed
 * for its lifetime (registers / two small LDS tables), ~40 % fewer instructions per sample.
 *
 * Bin indices: 512 bins need 9 bits.  The low 8 bits go out like the 1024-point path's (one dword = 4 consecutive spectra of
 * a column), the 9th as one bit per spectrum in a dword per (tile, column): 1.125 B per sample instead of 2.
 *
 * THE LOOP IS SKEWED (DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8, "C5, round 4", has the measurement behind every choice).  A spectrum's blocks make a round
 * trip store -> L2 -> cluster barrier -> load; with the loop in program order all eight waves of a CU sat through it (132 of 326 us).
 * Instead the same threads run stage A of spectrum u + 1 meanwhile: its first pass between the stores of spectrum u and their
 * s_waitcnt vmcnt(0), its wave-internal transpose and pass-2 twiddle products between the arrival at the cluster barrier and
 * everybody else's, its pass-2 butterflies beside the loads of the intermediate.  What the in-order return of a wave's loads
 * dictates around it:
 *   - the fp16 IQ of spectrum u + 3 is requested (LDS-DMA, two 32 KiB buffers) only once the loads of the intermediate have been
 *     used: a request to HBM ahead of them would delay them;
 *   - that LDS-DMA is issued by hand (inline asm): the compiler parks every barrier and LDS read that follows an LDS-DMA it knows
 *     about behind s_waitcnt vmcnt(0); the reads of the buffer sit behind an explicit vmcnt(0) of the requesting wave + a barrier;
 *   - work-group barriers are s_waitcnt lgkmcnt(0) + s_barrier (wg_barrier_lds): __syncthreads() would drain vmcnt;
 *   - a poll of a cluster counter through the vector path is a load too (it returns behind whatever its wave has in flight): the "has
 *     everybody read the intermediate" question therefore goes through the SCALAR path, every wave for itself (K1H_SPOLL, round 6; until
 *     round 5 the last wave, which requested no IQ, asked ahead of its epilogue's stores and a barrier passed the answer on);
 *   - the exact path's threshold table sits in LDS (ds_read has its own counter). */

/* Loads return IN ORDER and the first stage of a radix-16 pass pairs inputs j and j + 8: requested in this order, a butterfly's two inputs
 * arrive together (requested 0..15, the first butterfly waited for nine loads).  K1H_LOAD_ORDER=0: plain order (A/B builds). */
#ifndef K1H_LOAD_ORDER
#define K1H_LOAD_ORDER 1
#endif
#ifndef K1H_SC
#define K1H_SC false		/* (A/B builds) true: (2, 2) and the uniform twiddles of the first pass in scalar registers, like the 8192-point kernel */
#endif
#define K1H_PAIR(i) (K1H_LOAD_ORDER ? ((((i) & 1) << 3) | ((i) >> 1)) : (i))
#ifndef K1H_SPLIT
#define K1H_SPLIT 1			/* where the second pass of the NEXT spectrum's stage A runs (A/B builds): 0 stages A, B behind the arrival at the cluster
					 * barrier and C, D beside the loads of the intermediate; 1 all of it beside the loads; 2 all of it behind the arrival */
#endif
/* K1H_TIMING=1 (probe builds only, tools/k1h_phase_timing.py): s_memtime stamps per phase of the 65536-point kernel's loop, accumulated per
 * wave (waves 0, 3 and 7 of a work-group) into K1Params::dbg[(work-group * 3 + slot) * 16 + phase]. */
#ifndef K1H_TIMING
#define K1H_TIMING 0
#endif
/* K1H_SPOLL=1 (round 6): the "has every member read the intermediate" question is asked by EVERY wave for itself, through the SCALAR data
 * path (s_dcache_inv + s_load_dword: its answer does not queue behind the wave's vector stores -- another counter, another path), right before the
 * wave's stores of the next spectrum: by then the answer has long been yes.  Before, the last wave asked ahead of its epilogue (a vector
 * load behind the epilogue's stores would have waited for them), saw the spread between the cluster's members (~1 900 cycles per spectrum,
 * K1H_TIMING builds) and everybody else sat at a work-group barrier for the answer.  That barrier goes with it: nothing else needs it
 * (the exchange array's readers are separated from its next writers by the barrier behind the stores). */
#ifndef K1H_SPOLL
#define K1H_SPOLL 1
#endif
/* K1H_PRIO=1 (round 6): the YOUNGER wave of each SIMD (waves 4-7 of the work-group) issues at a higher priority than the older one.  Left to the
 * arbiter's age order the older wave of a SIMD won every tie and the younger ones reached each of the four barriers ~2 000 cycles late
 * (K1H_TIMING builds); with the priority the other way round the halves of the work-group take turns at being early -- the early wave's stores
 * and first pass run beside the late wave's epilogue -- and the kernel alone went from 231 to 216 us (profiles/r06_c5.md: the mirror image,
 * priority to the OLDER half, changes nothing; levels 1 / 2 / 3 measure the same). */
#ifndef K1H_PRIO
#define K1H_PRIO 1
#endif
static __device__ __forceinline__ uint32_t sload_fresh(const uint32_t *p)
{
	uint32_t v;
	/* (NOT `s_load_dword ... glc`: tools/ubench/poll_latency.hip measured 16 200 ticks per look for it on gfx950, against 217 for an
	 * invalidate of the scalar cache followed by a plain scalar load and 253 for a vector load with sc1) */
	asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
	return v;
}
#if K1H_TIMING
#define K1H_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
		const long long _now = __builtin_readcyclecounter(); hacc[i] += _now - hprev; hprev = _now; } while (0)
#else
#define K1H_STAMP(i) do { } while (0)
#endif

constexpr int kXaWave = 4 * 272;		/* stage-A exchange, elements per wave: [residue 4][jj 16][a 16], rows padded to 17 */
constexpr int kThrMax  = 520;			/* exact-bin thresholds kept in LDS (n_bins + 1 <= 513 doubles) */
constexpr int kTwRow = 9;			/* LDS twiddle tables: 8 twiddles per row, rows padded to 9 entries (72 B: 16 / 32 rows fall into different banks) */
/* The work-group's size decides the cluster's: a work-group of NWV waves takes 4 NWV residues (stage A) / offsets (stage B) of a spectrum,
 * so 64 / NWV work-groups make a cluster.  NWV = 8 is what runs: one work-group per CU, clusters of 8.  NWV = 4 -- TWO work-groups per CU,
 * members of different clusters of 16, so that one's arithmetic could run beside the other's LDS / memory phases -- was built in round 6,
 * parity-green, and 45-85 % SLOWER (450 against 245 us per frame: twice the members to wait for at each of a spectrum's two hand-overs,
 * and the hand-overs are what the loop's time is made of; profiles/r06_c5.md).  The geometry stays parametrised; only NWV = 8 is instantiated. */
template <int NWV> struct K1hGeom {
	static constexpr int kMem   = 64 / NWV;		/* members of a cluster */
	static constexpr int kRpm   = 4 * NWV;		/* residues = offsets per member */
	static constexpr int kXbLen = kRpm * 257;	/* stage-B exchange: [offset][jj3 16][a3 16], offsets padded to 257 */
	static constexpr int kXLen  = NWV * kXaWave > kXbLen ? NWV * kXaWave : kXbLen;	/* the two exchanges share one region (a barrier separates their uses) */
	static constexpr int kInLen = 256 * kRpm;	/* staged fp16 input of one spectrum: [row m 256][residue] dwords, 16-byte pieces permuted inside a row */
	static constexpr size_t kLds = ((size_t)kXLen + 16 * kTwRow + kRpm * kTwRow) * sizeof(float2) + (size_t)2 * kInLen * sizeof(uint32_t) + (size_t)kThrMax * sizeof(double);
						/* (exchange, two twiddle tables, staged input, thresholds) */
};

template <bool HALF, bool WRITE_FFT, int NWV>
__global__ __launch_bounds__(64 * NWV, 2)
void k1h_fused(const K1Params p)
{
	constexpr int N = 65536;
	typedef K1hGeom<NWV> G;
	constexpr int NT = 64 * NWV, kMem = G::kMem, kRpm = G::kRpm, kXLen = G::kXLen, kInLen = G::kInLen;
	/* Every wait on another work-group is bounded (a poll is ~1 us: seconds, far beyond any legitimate wait): a protocol failure
	 * ends the kernel with an error word the host turns into -EIO, it does not hang the GPU. */
	constexpr uint32_t kSpinLimit = 4u << 20;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	v2f *xa_all = reinterpret_cast<v2f *>(smem_raw);		/* stage A: one private exchange region per wave ... */
	v2f *xb = xa_all;						/* ... stage B: the work-group's exchange array, in the same memory */
	v2f *twa_t = xa_all + kXLen;					/* pass-2 twiddles [k2 16][8 of kTwRow] */
	v2f *tw3_t = twa_t + 16 * kTwRow;				/* pass-3 twiddles of this member's 32 offsets [32][8 of kTwRow] */
	uint32_t *inb = reinterpret_cast<uint32_t *>(tw3_t + kRpm * kTwRow);	/* fp16 IQ of the next two spectra (two buffers of kInLen dwords) */
	/* the exact-bin thresholds: the rare path that consults them must not wait for the loads and stores in flight (LDS reads have
	 * their own counter) */
	typedef const __attribute__((address_space(3))) double *lds_cdp;
	double *thr_g = reinterpret_cast<double *>(inb + 2 * kInLen);
	const lds_cdp thr_l = (lds_cdp)thr_g;

	const int tid = threadIdx.x;
	/* Cluster formation.  A work-group takes a ticket from the counter of the XCD it actually runs on (XCC_ID):
	 * tickets 8c .. 8c + 7 of an XCD are cluster c of that XCD, whatever the dispatcher did.  A cluster works once
	 * its 8 members are resident; complete clusters claim tiles until none is left, and a cluster still forming
	 * when the tiles run out (another kernel holds the CUs its members need) is abandoned as a whole: progress
	 * never depends on a work-group that is not resident. */
	__shared__ int sh_ticket, sh_tile;
	uint32_t xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	xcc &= 7;
	uint32_t *next_tile = p.sync + 63 * 64 + 56;		/* (on the last cluster's line; tickets sit at word 48) */
	const int ntiles = p.total / p.tile;
	if (tid == 0) {
		uint32_t *tick = p.sync + xcc * 8 * 64 + 48;		/* on the line of the XCD's first cluster */
		const uint32_t tk = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		int ok = 0;
		if (tk < 64) {
			/* the cluster's state: 0 forming, 1 complete (set by the holder of its 8th ticket: all 8 are resident),
			 * 2 abandoned (set by a member that saw the tiles run out first) -- one compare-and-swap decides */
			uint32_t *state = p.sync + ((int)xcc * 8 + (int)(tk / kMem)) * 64 + 24;
			uint32_t st = 0;
			if ((tk % kMem) == kMem - 1) {
				__hip_atomic_compare_exchange_strong(state, &st, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			} else {
				uint32_t spins = 0;
				while ((st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
					const bool tiles_left = (int)__hip_atomic_load(next_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ntiles;
					if (!tiles_left || ++spins > kSpinLimit) {	/* (a cluster that never fills is abandoned, never waited for) */
						if (tiles_left)
							*p.sync_err = 0x80000004u;	/* ... but with work left that is a failed call, not a quiet exit */
						uint32_t expect = 0;
						__hip_atomic_compare_exchange_strong(state, &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					}
					__builtin_amdgcn_s_sleep(8);
				}
			}
			ok = (st == 1);
		}
		sh_ticket = ok ? (int)tk : -1;
	}
	__syncthreads();
	/* The counters reset themselves: the last work-group to leave the kernel (an exit ticket, drawn behind everything else a work-group
	 * does with them) zeroes the whole array for the next launch -- no memset queued per frame (4.6 us each on this runtime). */
	auto leave = [&]() {
		__syncthreads();
		if (tid == 0)
			sh_ticket = (int)__hip_atomic_fetch_add(p.sync + 63 * 64 + 60, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__syncthreads();
		if (sh_ticket == (int)gridDim.x - 1)
			for (int e = tid; e < 64 * 64; e += NT)
				__hip_atomic_store(p.sync + e, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	if (sh_ticket < 0) {
		leave();
		return;
	}
	/* (everything that is the same for the whole work-group is forced into SGPRs: addresses are then a scalar base plus ONE
	 * 32-bit per-lane offset -- global_load / global_store ... s[base:base+1] -- instead of a 64-bit vector add per access) */
	const int ticket = __builtin_amdgcn_readfirstlane(sh_ticket);
	const int member = ticket % kMem;
	const int gc = (int)xcc * 8 + ticket / kMem;			/* cluster: up to 8 per XCD */
	uint32_t *c_a = p.sync + gc * 64;				/* stage A done */
	uint32_t *c_t = p.sync + gc * 64 + 16;				/* (round << 20) | tile, published by member 0 */
	uint32_t *c_b = p.sync + gc * 64 + 32;				/* stage B has read the intermediate */
	v2f *wint = reinterpret_cast<v2f *>(p.scratch) + (size_t)gc * N;	/* the cluster's intermediate: [offset / 32][residue][offset % 32] */
	const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wint);

	const int lane = tid & 63;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const v2f *twg = reinterpret_cast<const v2f *>(p.tw);
	const v2f two = { 2.0f, 2.0f };
	const BinConst bk = { p.binA, p.binC, p.amb, p.kappa, p.n_bins, p.thr };
	const float vmax_init = -1000.0f / F_HALF_LOG10_2;
	const float top = (float)(bk.nb - 1);

	/* ---- per-thread constants -------------------------------------------------------------------------------------
	 * stage A: residue q = 32 member + 4 wave + (lane >> 4); pass-1 item a = lane & 15 reads m = a + 16 j; after the
	 *          exchange the same lane is pass-2 item k2 = lane & 15 (twiddle index k2)
	 * stage B: offset kk = 32 member + (tid & 31); pass-3 item a3 = tid >> 5 reads residues q = a3 + 16 j3 (twiddle
	 *          index kk); after the exchange the same thread is pass-4 item jj3 = tid >> 5 (twiddle index kk + 256 jj3)
	 *          and owns columns kk + 256 jj3 + 4096 jj4 */
	const int sa = lane >> 4, ia = lane & 15;
	const int qa = kRpm * member + 4 * wv + sa;
	const int kkl = tid % kRpm, ib = tid / kRpm;
	const int kk = kRpm * member + kkl;
	const int col0 = kk + 256 * ib;
	const unsigned ucol0 = (unsigned)col0;				/* the one per-lane offset of every output access */
	/* the intermediate is [offset / kRpm][residue][offset % kRpm].  32 offsets per block: see the stores below; 16: a row is one 128-byte run */
	const unsigned wst0 = kRpm == 32 ? 8u * (unsigned)(qa * 32 + (ia ^ ((qa & 1) << 4)))		/* stage-A stores of even / odd jj (byte offsets) */
	                                 : 8u * (unsigned)(qa * 16 + ia);
	const unsigned wst1 = wst0 ^ 128u;
	const unsigned wld = kRpm == 32 ? 8u * (unsigned)(ib * 32 + (kkl ^ ((ib & 1) << 4)))		/* stage-B loads */
	                                : 8u * (unsigned)(ib * 16 + kkl);
	const __amdgpu_buffer_rsrc_t rs_wf = make_rsrc(p.wf), rs_part = make_rsrc(p.partial);

	const v2f w16c = twg[p.tw_off[0]], w8c = twg[p.tw_off[0] + 1], w163c = twg[p.tw_off[0] + 2];	/* W16, W8, W16^3: the first pass */
	v2f wab[HALF ? 8 : 1];						/* wab[j]: the window taps of this thread's pass-1 inputs j and j + 8, the pair of
									 * a stage-A butterfly (fp32 IQ, not a BASELINE configuration at this length: read
									 * where they are used -- its 32 staging registers leave no room for them) */
#pragma unroll
	for (int j = 0; j < (HALF ? 8 : 1); j++)
		wab[j] = v2f{ p.win[qa + 256 * (ia + 16 * j)], p.win[qa + 256 * (ia + 16 * (j + 8))] };
	v2f tw4[8];							/* pass 4: w^8, w^4, w^2, w^2 W8, w, w W16, w W8, w W16^3 of k = kk + 256 ib */
#pragma unroll
	for (int j = 0; j < 8; j++)
		tw4[j] = twg[p.tw_off[3] + (kk + 256 * ib) * 8 + j];
	for (int e = tid; e <= p.n_bins && e < kThrMax; e += NT)
		thr_g[e] = p.thr[e];
	for (int e = tid; e < 16 * 8; e += NT)
		twa_t[(e >> 3) * kTwRow + (e & 7)] = twg[p.tw_off[1] + e];
	for (int e = tid; e < kRpm * 8; e += NT)
		tw3_t[(e >> 3) * kTwRow + (e & 7)] = twg[p.tw_off[2] + (kRpm * member) * 8 + e];
	__syncthreads();

	v2f *xa = xa_all + wv * kXaWave;
	const int ea_w = sa * 272 + ia;			/* + 17 jj : pass-1 outputs [residue][jj][a] */
	const int ea_r = sa * 272 + ia * 17;		/* + j2    : pass-2 inputs of item k2 = ia */
	const int eb_w = kkl * 257 + ib;		/* + 16 jj3: pass-3 outputs [offset][jj3][a3] */
	const int eb_r = kkl * 257 + ib * 16;		/* + j4    : pass-4 inputs of item jj3 = ib */

	if (K1H_PRIO && wv >= NWV / 2)
		__builtin_amdgcn_s_setprio(2);
	uint32_t done = 0;						/* spectra this cluster has finished */
	uint32_t round = 0;						/* tiles this cluster has taken */

	typedef _Float16 h2 __attribute__((ext_vector_type(2)));
	/* fp16 IQ: the work-group fetches its 32 residues of a spectrum as whole 128-byte runs STRAIGHT INTO LDS (buffer_load_dwordx4 ... lds:
	 * no staging registers, no ds_write pass) -- one wave-instruction lands 64 x 16 B = 8 rows x 128 B back to back, so rows cannot be
	 * padded; the 16-byte piece pc of row m sits at slot 8 m + (pc ^ (m & 7)) instead (the permutation is applied to the per-lane SOURCE
	 * address and again to the read address; a wave's reads meet two-way conflicts at most).  A wave then finds the rows of its four
	 * residues in LDS (a wave gathering its own 16-byte pieces straight from memory touches every line eight times over: measured
	 * +205 us per frame against +37).  Two buffers: spectrum u of a tile in buffer u & 1.
	 * fp32 IQ (not a BASELINE configuration at this length) is gathered per lane where it is used. */
	/* (16 residues per member: rows of 64 B, one wave-instruction lands 16 of them; piece pc of row m at slot 4 m + (pc ^ ((m >> 2) & 3)):
	 * the 64 lanes of a read -- 16 rows x the 4 dwords of one piece -- then fall into 64 different banks) */
	constexpr int kRowsPerDma = 256 / kRpm;			/* rows one wave-instruction lands: 8 / 16 */
	const uint32_t iq_vo = kRpm == 32 ? 1024u * (unsigned)(lane >> 3) + 16u * (unsigned)((lane & 7) ^ ((lane >> 3) & 7))
	                                  : 1024u * (unsigned)(lane >> 2) + 16u * (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
	const int in_rd  = kRpm == 32 ? ia * 32 + ((wv ^ (ia & 7)) << 2) + sa	/* + kRpm * 16 j: row m = ia + 16 j, residue 4 wave + sa (dwords) */
	                              : ia * 16 + ((wv ^ ((ia >> 2) & 3)) << 2) + sa;
	const uint32_t inb_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)inb;
	if (PROBE_K1H(p) & 2)			/* (measurement only: noise-like input that is never loaded) */
		for (int e = tid; e < 2 * kInLen; e += NT)
			inb[e] = ((0x211fu + 977u * e) & 0x3fffu) | 0x20000000u | (((0x2c11u + 131u * e) & 0x3fffu) << 16) | ((e & 1u) << 15) | ((e & 2u) << 30);
	auto fetch_iq = [&](int t, int buf) {		/* row groups g = wave, wave + NWV - 1, ... (256 dwords each) into buffer `buf`; the last wave
							 * requests nothing: it polls the cluster counters, and a poll returns behind whatever
							 * its wave has in flight */
		constexpr int kFetchWaves = K1H_SPOLL ? NWV : NWV - 1;	/* (K1H_SPOLL: nobody polls through the vector path, every wave fetches) */
		if (!HALF || (PROBE_K1H(p) & 2) || wv >= kFetchWaves)
			return;
		if (PROBE_K1H(p) & 32) t = gc;	/* (measurement only: the same rows again and again) */
		/* global_load_lds_dwordx4 by hand: the compiler parks every LDS read and every __syncthreads() that follows an LDS-DMA it
		 * knows about behind s_waitcnt vmcnt(0) -- the request would be waited for at the very next barrier instead of an iteration
		 * later.  Whoever reads the buffer is behind an explicit `s_waitcnt vmcnt(..)` of the requesting wave and a barrier. */
		const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (size_t)t * p.hop + kRpm * member;
#pragma unroll 1
		for (int g = wv; g < kRpm; g += kFetchWaves) {
			const uint32_t *sk = src + 256 * kRowsPerDma * g;				/* 8 / 16 rows of 1 KiB */
			const uint32_t la = inb_lds + 4u * (unsigned)(buf * kInLen + 256 * g);
			uint32_t keep;
			asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 " K1H_IQ_MOD "\n\ts_mov_b32 m0, %0"
			             : "=&s"(keep) : "v"(iq_vo), "s"(sk), "s"(la) : "memory");
		}
	};

	v2f ra[16];				/* stage A of the spectrum AFTER the one stage B is working on */
	/* pass 1 (p = 1: no twiddles) of spectrum t and the 16 x 16 transpose inside the wave */
	auto stage_a1 = [&](int t, int buf) {
		if (PROBE_K1H(p) & 64) return;		/* (measurement only: stage B alone) */
		const __amdgpu_buffer_rsrc_t rs_f = make_rsrc(p.iq + (size_t)t * p.hop);
#pragma unroll
		for (int jo = 0; jo < 16; jo++) {
			const int j = K1H_PAIR(jo);
			v2f xv;
			if (HALF) {
				const uint32_t raw = inb[buf * kInLen + in_rd + 16 * kRpm * j];
				const h2 h = __builtin_bit_cast(h2, raw);
				xv = v2f{ (float)h.x, (float)h.y };		/* v_cvt_f32_f16: exact */
			} else if (PROBE_K1H(p) & 2) {
				xv = v2f{ 0.01f * (float)(((tid * 37 + j * 11) & 63) - 32), 0.01f * (float)(((tid * 29 + j * 7) & 63) - 31) };
			} else {
				xv = bld_v2f<kAuxNT>(rs_f, 8u * (unsigned)(qa + 256 * ia), 32768u * j);
			}
			ra[j] = xv;
		}
		/* first pass (p = 1), the window of fft.cl:415-417 on its stage-A butterflies */
		if constexpr (HALF) {
			pass16_first<K1H_SC, false>(ra, wab, w16c, w8c, w163c, two);
		} else {
			v2f wl[8];
#pragma unroll
			for (int j = 0; j < 8; j++)
				wl[j] = v2f{ p.win[qa + 256 * (ia + 16 * j)], p.win[qa + 256 * (ia + 16 * (j + 8))] };
			pass16_first<K1H_SC, false>(ra, wl, w16c, w8c, w163c, two);
		}
	};
	/* ... and the 16 x 16 transpose inside the wave that follows it */
	auto stage_a1x = [&]() {
		if (PROBE_K1H(p) & 64) return;
#pragma unroll
		for (int jj = 0; jj < 16; jj++)
			xa[ea_w + 17 * jj] = ra[R16_PERM(jj)];
		wave_lds_sync();
#pragma unroll
		for (int jo = 0; jo < 16; jo++)
			ra[K1H_PAIR(jo)] = xa[ea_r + K1H_PAIR(jo)];
		wave_lds_sync();
	};
	/* pass 2, p = 16, k = ia */
#ifndef K1H_TW_REGS
#define K1H_TW_REGS 0		/* 1: the pass-2 / pass-3 twiddles of a thread (fixed for its lifetime) in registers instead of 16 LDS reads per spectrum.
				 * Measured (round 5): the kernel ALONE 2.6 % faster (15 850 against 16 270 cycles per spectrum, K1H_TIMING builds) -- and the
				 * path 10 % slower (209 against 231 GSamples/s, three interleaved runs each): 233 instead of 209 VGPRs leave the scan / merge
				 * kernels of the previous frame no registers on a CU this kernel occupies, and the frame's tail no longer runs beside it */
#endif
#if K1H_TW_REGS
	v2f twa_r[8], tw3_r[8];
#pragma unroll
	for (int j = 0; j < 8; j++) {
		twa_r[j] = twa_t[ia * kTwRow + j];
		tw3_r[j] = tw3_t[kkl * kTwRow + j];
	}
#else
	const v2f *twa_r = twa_t + ia * kTwRow;
	const v2f *tw3_r = tw3_t + kkl * kTwRow;
#endif
	auto stage_a2_ab = [&]() { if (PROBE_K1H(p) & 64) return; pass16_ab<K1H_SC, false>(ra, twa_r[0], twa_r[1], two); };
	auto stage_a2_cd = [&]() { if (PROBE_K1H(p) & 64) return; pass16_cd<K1H_SC, false>(ra, twa_r[2], twa_r[3], twa_r[4], twa_r[5], twa_r[6], twa_r[7], two); };
	auto stage_a2 = [&]() { stage_a2_ab(); stage_a2_cd(); };

#if K1H_TIMING
	long long hacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	long long hprev = __builtin_readcyclecounter();
#endif
	uint32_t *bins_lo = p.bins;					/* [total / 4][N] dwords: 4 spectra x low 8 bits */
	uint32_t *bins_hi = p.bins + (size_t)(p.total >> 2) * N;	/* [total / tile][N] dwords: bit u = 9th bit of the tile's spectrum u */

	for (;;) {
	/* member 0 claims the cluster's next tile */
	if (tid == 0) {
		uint32_t v;
		if (member == 0) {
			v = __hip_atomic_fetch_add(next_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (v > 0xfffffu) v = 0xfffffu;
			__hip_atomic_store(c_t, ((round + 1) << 20) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			uint32_t spins = 0;
			while (((v = __hip_atomic_load(c_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 20) != round + 1) {
				if (++spins > kSpinLimit) { *p.sync_err = 0x80000001u; v = 0xfffffu; break; }	/* fail the call, not the GPU */
				__builtin_amdgcn_s_sleep(2);
			}
			v &= 0xfffffu;
		}
		sh_tile = (int)v;
	}
	__syncthreads();				/* (also: every read of the exchange array by the previous tile's last spectrum is done) */
	const int tile = __builtin_amdgcn_readfirstlane(sh_tile);
	round++;
	if (tile >= ntiles)
		break;
	const int t0 = tile * p.tile;
	float live[16], vmax[16];
	uint32_t plo[16], phi[16];
#pragma unroll
	for (int c = 0; c < 16; c++) { live[c] = 0.0f; vmax[c] = vmax_init; plo[c] = 0; phi[c] = 0; }
	auto epilogue = [&](v2f (&r)[16], const int t, const int u) {
		if (WRITE_FFT) {
#pragma unroll
			for (int c = 0; c < 16; c++)
				bst_v2f<0>(r[R16_PERM(c)], make_rsrc(reinterpret_cast<v2f *>(p.fft_out) + (size_t)t * N), 8u * ucol0, 32768u * c);
		}

		/* epilogue (display.cl:136-150,161-168), 9-bit bin indices: low byte into the quad's dword, 9th bit into the tile's */
		const bool store_row = (t >= p.wf_first) && !(PROBE_K1H(p) & 4);
		const uint32_t wf_so = (uint32_t)((p.wf_pos0 + t) & p.wf_mask) * (uint32_t)(N * 4);
		const int sh8 = 8 * (u & 3);
		/* four samples at a time: fast path, ONE branch for the four (rare: some sample is not provably exact -- find it again and
		 * decide it against the exact thresholds), then the updates and stores */
#pragma unroll
		for (int g = 0; g < 4; g++) {
			float l2g[4]; uint32_t bng[4]; uint32_t amb = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const v2f x = r[R16_PERM(4 * g + k)];
				uint32_t ab;
				const float rr = bin_fast(x.x, x.y, bk, &l2g[k], &ab);
				amb = amb > ab ? amb : ab;		/* v_max_u32: NaN / inf order above every finite measure */
				bng[k] = (uint32_t)(int)__builtin_amdgcn_fmed3f(rr, 0.0f, top);
			}
			if (amb > __float_as_uint(bk.amb) && !(PROBE_K1H(p) & 16)) {
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const v2f x = r[R16_PERM(4 * g + k)];
					const float v = __builtin_fmaf(bk.A, l2g[k], bk.C);
					const float a = __builtin_fmaf(__builtin_fabsf(l2g[k]), bk.kappa, __builtin_fabsf(v - __builtin_rintf(v)));
					if (!(a <= bk.amb)) {
						float nl2;
						bng[k] = bin_exact(x.x, x.y, l2g[k], (int)bng[k], thr_l, bk.nb, &nl2);
						l2g[k] = nl2;
					}
				}
			}
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int c = 4 * g + k;
				const uint32_t bn = bng[k];
				const float l2v = l2g[k];
				plo[c] |= (bn & 0xffu) << sh8;
				phi[c] |= (bn >> 8) << u;
				live[c] = __builtin_fmaf(live[c], p.w, l2v);
				vmax[c] = max_f32(vmax[c], l2v);
				/* rows and bin indices are streamed out non-temporally: plain stores allocate in the XCD's L2 and push the cluster's
				 * intermediate out of it */
				if (store_row)
					__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l2v * F_HALF_LOG10_2), rs_wf, 4u * ucol0, wf_so + 16384u * c, K1H_OUT_AUX);
			}
		}
		if ((u & 3) == 3 && !(PROBE_K1H(p) & 4)) {
			const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(bins_lo + (size_t)(t >> 2) * N);
#pragma unroll
			for (int c = 0; c < 16; c++) {
				__builtin_amdgcn_raw_buffer_store_b32(plo[c], rs_lo, 4u * ucol0, 16384u * c, K1H_OUT_AUX);
				plo[c] = 0;
			}
		}
	};
	/* the tile's first spectrum: nothing to hide its stage A behind.  Input buffers: spectrum u of the tile in buffer u & 1 */
	fetch_iq(t0, 0);
	if (1 < p.tile)
		fetch_iq(t0 + 1, 1);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	wg_barrier_lds();
	stage_a1(t0, 0);
	stage_a1x();
	wg_barrier_lds();				/* every wave has its rows out of buffer 0 */
	if (2 < p.tile)
		fetch_iq(t0 + 2, 0);
	stage_a2();

	/* The loop is skewed: while spectrum u's blocks travel to the L2 (stores), to the other members (cluster wait) and back (loads),
	 * the same threads run stage A of spectrum u + 1 -- its first pass between the stores and the arrival at the cluster barrier,
	 * its second between the loads of the intermediate and their use. */
#pragma unroll 1
	for (int u = 0; u < p.tile; u++) {
		const int t = t0 + u;
		const bool more = (u + 1 < p.tile);

		K1H_STAMP(0);		/* loop overhead, tile claim (first spectrum of a tile) */
		/* (every member has read the previous spectrum out of the intermediate: the last wave looked before its epilogue)
		 * every read of the stage-B exchange array is done -- stage A writes the same memory */
		if (K1H_SPOLL) {
			/* every member has read the previous spectrum out of the intermediate? */
			if (!(PROBE_K1H(p) & 1)) {
				uint32_t spins = 0;
				while ((int)(sload_fresh(c_b) - (uint32_t)kMem * done) < 0) {
					if (++spins > kSpinLimit) { if (lane == 0) *p.sync_err = 0x80000002u; break; }
					__builtin_amdgcn_s_sleep(1);
				}
			}
		} else {
			wg_barrier_lds();
		}
		K1H_STAMP(1);		/* top barrier: waiting for the work-group's slowest wave (K1H_SPOLL: this wave's own look at the counter) */
		if (!(PROBE_K1H(p) & (8 | 512))) {
			/* w[256 q + kk], kk = ia + 16 jj2, at [kk >> 5][q][(kk & 31) ^ 16 (q & 1)]: 16 lanes x 8 B = 128-byte runs; odd residues
			 * keep their two halves swapped so that one store instruction (one jj for every lane) is spread over both halves of the
			 * 256-byte rows -- both values of the address bit that picks an L2 channel -- instead of one */
#pragma unroll
			for (int jj = 0; jj < 16; jj++) {
				if (kRpm == 32) bst_v2f<0>(ra[R16_PERM(jj)], rs_w, (jj & 1) ? wst1 : wst0, 65536u * (jj >> 1));
				else            bst_v2f<0>(ra[R16_PERM(jj)], rs_w, wst0, 32768u * jj);	/* (one instruction: four residues = 512 B in a row) */
			}
		}
		if (more)
			stage_a1(t + 1, (u + 1) & 1);			/* (while the stores travel) */
		K1H_STAMP(2);		/* intermediate stores issued + first pass of the next spectrum */
		/* this wave's blocks are in the L2 (and the input rows it requested most of an iteration ago in LDS) */
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		K1H_STAMP(3);		/* waiting for the stores' acknowledgements (and the IQ requested an iteration ago) */
		wg_barrier_lds();
		if (tid == 0)
			__hip_atomic_fetch_add(c_a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		K1H_STAMP(4);		/* barrier + arrival */
		if (more) {
			if (K1H_SPLIT != 3)
				stage_a1x();				/* (while the arrivals travel) */
			if (K1H_SPLIT == 0)
				stage_a2_ab();				/* second pass, stages A and B (C and D: beside the loads below) */
			else if (K1H_SPLIT == 2)
				stage_a2();
		}

		K1H_STAMP(5);		/* transpose (+ what of the second pass runs here) */
		if (tid == 0 && !(PROBE_K1H(p) & 1)) {
			uint32_t spins = 0;
			while ((int)(__hip_atomic_load(c_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (uint32_t)kMem * (done + 1)) < 0) {
				if (++spins > kSpinLimit) { *p.sync_err = 0x80000003u; break; }
				__builtin_amdgcn_s_sleep(1);
			}
		}
		wg_barrier_lds();
		asm volatile("" ::: "memory");
		K1H_STAMP(6);		/* cluster barrier: poll + work-group barrier */

		/* ================= stage B: offsets kk = 32 member .. + 31 ================= */
		v2f r[16];
		if (!(PROBE_K1H(p) & (8 | 256))) {
			/* residues q = ib + 16 j3 (q & 1 = ib & 1); sc1: the loads miss the CU's L1 by construction and are served by the L2 */
#pragma unroll
			for (int jo = 0; jo < 16; jo++)
				r[K1H_PAIR(jo)] = bld_v2f<kAuxSC1>(rs_w, wld, (uint32_t)(2048 * kRpm) * member + (uint32_t)(128 * kRpm) * K1H_PAIR(jo));
		} else {
#pragma unroll
			for (int j = 0; j < 16; j++)
				r[j] = ra[j];
		}
		if (more) {						/* (while the loads travel) */
			if (K1H_SPLIT == 0)
				stage_a2_cd();
			else if (K1H_SPLIT == 1)
				stage_a2();
			else if (K1H_SPLIT == 3) {
				stage_a1x();
				stage_a2();
			}
		}
		K1H_STAMP(7);		/* loads of the intermediate issued + second pass of the next spectrum */
		if (!(PROBE_K1H(p) & 128))	/* (128, measurement only: stage A alone -- no stage-B arithmetic, exchange or epilogue; the barriers stay) */
		pass16_ab<K1H_SC, false>(r, tw3_r[0], tw3_r[1], two);				/* pass 3, p = 256, k = kk */
#if K1H_TIMING
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
		K1H_STAMP(8);		/* third pass, stages A and B: includes the wait for the loads */
		/* spectrum u + 3 is requested into the buffer spectrum u + 1 has been read out of by every wave (two barriers ago); it is
		 * waited for by the `vmcnt(0)` of the NEXT iteration.  Requested only now that the loads of the intermediate have been used:
		 * loads return in order, and these come from HBM */
		if (u + 3 < p.tile)
			fetch_iq(t + 3, (u + 1) & 1);
		if (!(PROBE_K1H(p) & 128)) {
		pass16_cd<K1H_SC, false>(r, tw3_r[2], tw3_r[3], tw3_r[4], tw3_r[5], tw3_r[6], tw3_r[7], two);
#pragma unroll
		for (int jj = 0; jj < 16; jj++)
			xb[eb_w + 16 * jj] = r[R16_PERM(jj)];
		} else {
			asm volatile("s_waitcnt vmcnt(0)" :: "v"(r[0]), "v"(r[15]) : "memory");
		}
		K1H_STAMP(9);		/* IQ request + third pass, stages C and D + exchange stores */
		wg_barrier_lds();
		if (tid == 0)							/* everybody's loads of the intermediate have landed */
			__hip_atomic_fetch_add(c_b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		done++;
		K1H_STAMP(10);		/* exchange barrier */
		if (!(PROBE_K1H(p) & 128)) {
#pragma unroll
		for (int jo = 0; jo < 16; jo++)
			r[K1H_PAIR(jo)] = xb[eb_r + K1H_PAIR(jo)];
		pass16_ab<K1H_SC, false>(r, tw4[0], tw4[1], two);					/* pass 4, p = 4096, k = kk + 256 ib */
		pass16_cd<K1H_SC, false>(r, tw4[2], tw4[3], tw4[4], tw4[5], tw4[6], tw4[7], two);
		}

		K1H_STAMP(11);		/* exchange loads + fourth pass */
		/* every member has read this spectrum out of the intermediate?  (they said so about a pass ago.)  Asked here because this
		 * wave has nothing in flight now: behind the epilogue's stores the answer would wait for them.  (Round 5, K1H_TIMING build: the
		 * ~2000 cycles this wave spends here per spectrum are the spread between the cluster's members, not a round trip -- requesting
		 * the counter one pass EARLIER and looking at the answer here returned "not yet" and cost 50 us per frame on top.) */
		if (!K1H_SPOLL && tid == NT - 64 && !(PROBE_K1H(p) & 1)) {
			uint32_t spins = 0;
			while ((int)(__hip_atomic_load(c_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (uint32_t)kMem * done) < 0) {
				if (++spins > kSpinLimit) { *p.sync_err = 0x80000002u; break; }
				__builtin_amdgcn_s_sleep(1);
			}
		}

		K1H_STAMP(12);		/* "everyone has read the intermediate" poll (last wave only) */
		if (!(PROBE_K1H(p) & 128))
		epilogue(r, t, u);
		K1H_STAMP(13);		/* epilogue */
	}
	if (!(PROBE_K1H(p) & 4)) {
		const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(bins_hi + (size_t)tile * N);
#pragma unroll
		for (int c = 0; c < 16; c++)
			__builtin_amdgcn_raw_buffer_store_b32(phi[c], rs_hi, 4u * ucol0, 16384u * c, K1H_OUT_AUX);
	}
#pragma unroll
	for (int c = 0; c < 16; c++)
		bst_v2f<0>(v2f{ live[c] * F_HALF_LOG10_2, (vmax[c] == vmax_init) ? -1000.0f : vmax[c] * F_HALF_LOG10_2 },
		           rs_part, 8u * ucol0, (uint32_t)tile * (uint32_t)(N * 8) + 32768u * c);
	}
#if K1H_TIMING
	if (p.dbg && lane == 0 && (wv == 0 || wv == NWV / 2 - 1 || wv == NWV - 1)) {
		const int slot = (wv == 0) ? 0 : (wv == NWV / 2 - 1) ? 1 : 2;
		for (int i = 0; i < 16; i++)
			p.dbg[((size_t)blockIdx.x * 3 + slot) * 16 + i] = hacc[i];
	}
#endif
	leave();
}

template <int NWV>
static hipError_t launch_k1h_form(const K1Params &p0, hipStream_t s)
{
	typedef void (*k1h_fn)(const K1Params);
	static const k1h_fn fn[4] = { k1h_fused<false, false, NWV>, k1h_fused<true, false, NWV>, k1h_fused<false, true, NWV>, k1h_fused<true, true, NWV> };
	constexpr size_t lds = K1hGeom<NWV>::kLds;
	/* (the attribute belongs to the function object of the CURRENT device: once per device, not once per process) */
	static unsigned long long attr_dev = 0;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
		return hipErrorInvalidDevice;
	if (!(attr_dev >> dev & 1)) {
		for (int i = 0; i < 4; i++) {
			const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn[i]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			if (e != hipSuccess)
				return e;
		}
		attr_dev |= 1ull << dev;
	}
	/* (the counters in p0.sync are zero: cleared at allocation, and by the last work-group of every launch) */
	/* 32 clusters: 8 work-groups of 8 waves, one per CU -- or 16 work-groups of 4 waves, two per CU */
	hipLaunchKernelGGL(fn[(p0.iq_half ? 1 : 0) | (p0.fft_out ? 2 : 0)], dim3(256 * 8 / NWV), dim3(64 * NWV), lds, s, p0);
	return hipGetLastError();
}

static hipError_t launch_k1h(const K1Params &p0, hipStream_t s)
{
	/* tiles of 4 .. 32 spectra (whole quads of low bytes, the 9th bits of a tile in one dword); tile index: 20 bits of the claim word */
	if (!p0.sync || !p0.scratch || p0.tile < 4 || p0.tile > 32 || (p0.tile & 3) || p0.total % p0.tile || p0.total / p0.tile >= (1 << 20))
		return hipErrorInvalidValue;
	return launch_k1h_form<8>(p0, s);
}


hipError_t launch_k1(const K1Params &p, hipStream_t s)
{
	if (p.variant == 4)
		return launch_k1h(p, s);
	const int tiles = p.total / p.tile;
	if (p.variant == 3) {
		static unsigned long long attr_dev = 0;		/* (the attribute belongs to the function object of the CURRENT device: once per device) */
		if (p.log2n == 10) {
			/* N = 1024 with 16-bit bin indices (more than 256 bins): the general kernel at 128 threads per spectrum */
			constexpr int N = 1024;
			constexpr int lds = (N + ((N / 2 - 8) / 7) * 7 + N / 2) * 8 + N * 4;		/* exchange slab + the reference's twiddles + window */
			const int blocks = tiles < 4096 ? tiles : 4096;
			if (p.fft_out)
				hipLaunchKernelGGL((k1big_fft_bin<10, true>), dim3(blocks), dim3(N / 8), lds, s, p);
			else
				hipLaunchKernelGGL((k1big_fft_bin<10, false>), dim3(blocks), dim3(N / 8), lds, s, p);
			return hipGetLastError();
		}
		if (p.log2n != 13 || (p.tile & 7) || (p.total & 7))	/* (the kernel packs the 9th bits of spectra 8 u .. 8 u + 7 of a tile into one byte per column) */
			return hipErrorInvalidValue;
		/* N = 8192: 16 points per thread, tables in registers, overlap reuse in registers (k1w_fft_bin); any hop */
		constexpr int ldsw = 2 * 8192 * 8 + 520 * 8;	/* two slabs + the exact-bin thresholds */
		typedef void (*k1w_fn)(const K1Params);
		static const k1w_fn fns[5] = { k1w_fft_bin<8>, k1w_fft_bin<4>, k1w_fft_bin<2>, k1w_fft_bin<1>, k1w_fft_bin<16> };
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
			return hipErrorInvalidDevice;
		if (!(attr_dev >> dev & 1)) {
			for (int i = 0; i < 5; i++) {
				hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fns[i]), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);
				if (e != hipSuccess)
					return e;
			}
			attr_dev |= 1ull << dev;
		}
		/* rows of 512 samples the next window of a tile shares with this one: hop = 8192 / R, R = 2, 4, 8, 16; any other hop: none */
		const int which = (p.hop == 4096) ? 0 : (p.hop == 2048) ? 1 : (p.hop == 1024) ? 2 : (p.hop == 512) ? 3 : 4;
		/* one 8-wave work-group per CU -- or per CU of the share the host leaves to this kernel (K1Params.cus: the count and
		 * merge kernels of the previous launch run on the rest) */
		const int all_cus = p.n_cus > 0 ? p.n_cus : 256;
		const int cus = (p.cus > 0 && p.cus < all_cus && tiles % p.cus == 0) ? p.cus : all_cus;
		const int bw = tiles < cus ? tiles : cus;
		hipLaunchKernelGGL(fns[which], dim3(bw), dim3(512), ldsw, s, p);
		return hipGetLastError();
	}
	if (p.variant == 2) {
		const int maxb = 256 * 2 * K1V2_WAVES_PER_SIMD;	/* resident 2-wave work-groups on 256 CUs */
		int blocks = tiles < maxb ? tiles : maxb;
		if (p.fft_out)
			hipLaunchKernelGGL(k1v2_fft_bin<true>, dim3(blocks), dim3(128), 0, s, p);
		else
			hipLaunchKernelGGL(k1v2_fft_bin<false>, dim3(blocks), dim3(128), 0, s, p);
		return hipGetLastError();
	}
	int blocks = (tiles + 3) / 4;
	/* FOSPHOR_AMD_K1_BLOCKS: debugging aid (e.g. 256 = one wave per SIMD, for phase timing) */
	static const int max_blocks = [] { const char *e = getenv("FOSPHOR_AMD_K1_BLOCKS"); const int v = e ? atoi(e) : 0;
	                                   return (v > 0 && v < kK1MaxBlocks) ? v : kK1MaxBlocks; }();
	if (blocks > max_blocks)
		blocks = max_blocks;		/* persistent: 2 work-groups per CU */
	if (p.fft_out)
		hipLaunchKernelGGL((k1_fft_bin<true, false>), dim3(blocks), dim3(256), 0, s, p);
	else
		if (p.n_bins == 256)
			hipLaunchKernelGGL((k1_fft_bin<false, true>), dim3(blocks), dim3(256), 0, s, p);
		else
			hipLaunchKernelGGL((k1_fft_bin<false, false>), dim3(blocks), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* Test hook: the K1 epilogue alone on FFT values read from memory */
__global__ __launch_bounds__(256)
void k_bin_hook(const float2 *__restrict__ fft, uint8_t *__restrict__ bin, float *__restrict__ pwr, int n,
                const K1Params p, int force_exact)
{
	const BinConst bk = { p.binA, p.binC, p.amb, p.kappa, p.n_bins, p.thr };
	const float top = (float)(bk.nb - 1);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		float2 v = fft[i];
		float l2; uint32_t ab;
		const float r = bin_fast(v.x, v.y, bk, &l2, &ab);
		/* the two ways the K1 variants turn the guess into an index: saturating byte convert
		 * (<= 256 bins) or clamp + integer convert (16-bit indices) */
		uint32_t b = p.bins16 ? (uint32_t)(int)__builtin_amdgcn_fmed3f(r, 0.0f, top) : (pack_bin(r, top, 0, 0) & 0xff);
		if (ab > __float_as_uint(bk.amb) || force_exact)
			b = bin_exact(v.x, v.y, l2, (int)__builtin_amdgcn_fmed3f(r, 0.0f, top), ThrScalar{ bk.thr }, bk.nb, &l2);
		if (p.bins16)
			reinterpret_cast<uint16_t *>(bin)[i] = (uint16_t)b;
		else
			bin[i] = (uint8_t)b;
		pwr[i] = l2 * F_HALF_LOG10_2;
	}
}

hipError_t launch_bin_hook(const float2 *fft, uint8_t *bin, float *pwr, int n,
                           const K1Params &p, int force_exact, hipStream_t s)
{
	int blocks = (n + 255) / 256;
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(k_bin_hook, dim3(blocks), dim3(256), 0, s, fft, bin, pwr, n, p, force_exact);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* K2: hit counts per (slab of 64 columns, chunk)                            */
/* ------------------------------------------------------------------------ */

/* One lane per column: the 64 lanes of a wave count 64 different columns, so an LDS atomic
 * never meets a bank conflict between different (bin, column) cells.  Columns c and c + 32 of
 * the slab share one dword (low / high 16 bits; a chunk has <= 1024 spectra, so the low half
 * cannot carry into the high one): the two lanes that meet in a bank are serialised by the
 * hardware either way, and the histogram is half the size (32 KiB at 256 bins), which lets a
 * work-group of this kernel sit beside the two K1 work-groups of a CU.
 * (The previous layout, display.cl:96,176's [bin][16] with 4 spectra x 16 columns per wave,
 * had 4 lanes per column in every atomic instruction and kept the LDS pipe busy ~3x longer.) */
/* Independent index loads in flight per thread (IF): 8 for the per-batch chunks of the 1024-point path (45 VGPRs: still beside two K1
 * waves of 228 on a SIMD; 4 -> 8: K2 59 -> 51 us beside K1, path +1.6 %; 12 / 16 = 61 / 63 VGPRs no longer fit there), 4 where a work-group
 * counts several chunks (sharded frames: 532 against 527 GSamples/s) and for the 16-bit-index geometries (N = 8192: 4 / 8 / 16 -> 62 / 67 /
 * 72 us). */
/* NW waves per work-group: 4 where the kernel has to fit beside K1 (8-bit indices, N = 1024); 16 for the
 * 16-bit-index geometries, whose grids are small (N/64 x chunks) and whose rows are latency-bound */
/* K2_TIMING (probe builds with -DFOSPHOR_AMD_PROBES -DK2_TIMING, tools/k2_phase_timing.py): s_memtime stamps per phase of the count kernel, summed
 * over the work-groups' first waves into g_k2_time[] (read through fosphor_amd_debug_k2_timing) */
#if defined(FOSPHOR_AMD_PROBES) && defined(K2_TIMING)
__device__ unsigned long long g_k2_time[16];
#define K2_STAMP(i) do { if (tid == 0) { const long long _n = __builtin_readcyclecounter(); atomicAdd(&g_k2_time[i], (unsigned long long)(_n - k2prev)); k2prev = _n; } } while (0)
#else
#define K2_STAMP(i) do { } while (0)
#endif
#ifndef K2_DBG
#define K2_DBG 0		/* measurement only (wrong counts): 1 no LDS atomics, 2 plain LDS stores instead (profiles/r04_ceiling.md) */
#endif
/* one hit: row `bin` of the [bin][32] histogram (128 bytes per row), the lane's column at byte hc4 */
static __device__ __forceinline__ void lds_count(uint32_t *h, uint32_t bin, uint32_t hc4, uint32_t inc)
{
	atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(h) + ((bin << 7) + hc4)), inc);	/* v_lshl_add_u32 + ds_add_u32 */
}
/* v & 0xffff / v & 0xff as ONE instruction the compiler cannot merge into the shift that follows */
static __device__ __forceinline__ uint32_t lo16(uint32_t v) { uint32_t r; asm("v_and_b32 %0, 0xffff, %1" : "=v"(r) : "v"(v)); return r; }
static __device__ __forceinline__ uint32_t lo8(uint32_t v)  { uint32_t r; asm("v_and_b32 %0, 0xff, %1" : "=v"(r) : "v"(v)); return r; }

template <int NW, int IF>
__global__ __launch_bounds__(64 * NW)
void k2_count(const K2Params p)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t h[];	/* [n_bins][32] packed pairs */
	__shared__ float red_s[NW][64], red_m[NW][64];

	const int tid  = threadIdx.x;
	const int lane = tid & 63;
	const int wv   = __builtin_amdgcn_readfirstlane(tid >> 6);	/* (the compiler does not know it is wave-uniform: with it in an SGPR the row
								 * addresses below are scalar arithmetic + one VGPR of lane offset; as a VGPR every
								 * load cost a v_mul_lo_u32 and a 64-bit add: 23 instead of 31 VGPRs, 2 instead of 22
								 * v_mul_lo_u32; the kernel's time did not change, it does not wait for its VALUs) */
	const int x0   = blockIdx.x * 64;
	const int c    = PROBE_SAME(p) ? 0 : blockIdx.y;	/* chunk index within the launch */
	const int cpb  = p.batch / p.chunk;		/* chunks per batch */
	const int f    = c / cpb;			/* batch index */
	const int t_in = (c - f * cpb) * p.chunk;	/* first spectrum of the chunk within its batch */
	const int nb   = p.n_bins;
	const int hcol = lane & 31;
	const uint32_t inc = (lane & 32) ? 0x10000u : 1u;
#if defined(FOSPHOR_AMD_PROBES) && defined(K2_TIMING)
	long long k2prev = __builtin_readcyclecounter();
	if (tid == 0) atomicAdd(&g_k2_time[15], 1ull);		/* work-groups */
#endif

	__shared__ uint32_t rowbits[16];		/* n_bins <= 512 */
	{
		/* 16 bytes per instruction (n_bins is a multiple of 16: nb * 32 dwords = whole uint4s) */
		uint4 *h4 = reinterpret_cast<uint4 *>(h);
		for (int i = tid; i < nb * 8; i += 64 * NW)
			h4[i] = make_uint4(0u, 0u, 0u, 0u);
	}
	if (tid < 16)
		rowbits[tid] = 0;
	K2_STAMP(0);		/* zeroing issued */
	__syncthreads();
	K2_STAMP(1);		/* barrier */

	/* bins: one dword = 4 consecutive spectra of one column (8-bit indices), or 2 (16-bit
	 * indices, n_bins > 256 or N > 1024); a wave reads 256 contiguous bytes per row */
	/* (uniform base pointer + 32-bit lane offsets: one address register per load in flight) */
	if (p.bins9) {
		/* 9-bit indices of the 65536-point kernel: a wave takes whole tiles -- one dword of 9th bits per lane and tile, then the
		 * tile's (at most 8) dwords of low bytes, all requested before the first is used */
		const uint32_t n = p.n, qpt = (uint32_t)p.tile >> 2, ntl = (uint32_t)(p.chunk / p.tile);
		const uint32_t *lo = p.bins + (size_t)c * (p.chunk >> 2) * n + x0 + lane;
		const uint32_t *hi = p.bins + (size_t)(p.total >> 2) * n + (size_t)c * ntl * n + x0 + lane;
#pragma unroll 1
		for (uint32_t tl = wv; tl < ntl; tl += NW) {
#ifndef K2_NT9
#define K2_NT9 1		/* (0: A/B builds) the index planes of N = 65536 read non-temporally (they are read once; the FFT kernel of the next frame keeps its
				 * intermediates in the same L2) */
#endif
			const uint32_t hv = K2_NT9 ? __builtin_nontemporal_load(&hi[(size_t)tl * n]) : hi[(size_t)tl * n];
			uint32_t v[8];
#pragma unroll
			for (uint32_t u = 0; u < 8; u++)
				v[u] = (u < qpt) ? (K2_NT9 ? __builtin_nontemporal_load(&lo[(size_t)(tl * qpt + u) * n]) : lo[(size_t)(tl * qpt + u) * n]) : 0u;
#pragma unroll
			for (uint32_t u = 0; u < 8; u++) {
				if (u < qpt) {			/* uniform */
					const uint32_t h4 = hv >> (4 * u);
					atomicAdd(&h[(((v[u]      ) & 0xff) | ((h4 & 1u) << 8)) * 32 + hcol], inc);
					atomicAdd(&h[(((v[u] >>  8) & 0xff) | ((h4 & 2u) << 7)) * 32 + hcol], inc);
					atomicAdd(&h[(((v[u] >> 16) & 0xff) | ((h4 & 4u) << 6)) * 32 + hcol], inc);
					atomicAdd(&h[(((v[u] >> 24)       ) | ((h4 & 8u) << 5)) * 32 + hcol], inc);
				}
			}
		}
	} else if (p.bins8p1) {
		/* the 8192-point kernel's indices: shorts [t / 2][column] (low bytes of two spectra) and, behind them, bytes [t / 8][column]
		 * (the 9th bits of eight).  A wave takes whole groups of eight spectra of its 64 columns: one byte and four shorts per lane,
		 * two groups' worth requested before the first is used.  Scalar descriptor + one lane offset + scalar row offsets. */
		const __amdgpu_buffer_rsrc_t rlo = make_rsrc(reinterpret_cast<const char *>(p.bins) + ((size_t)c * (p.chunk >> 1) * p.n + x0) * 2);
		const __amdgpu_buffer_rsrc_t rhi = make_rsrc(reinterpret_cast<const char *>(p.bins) + (size_t)p.total * p.n + (size_t)c * (p.chunk >> 3) * p.n + x0);
		const uint32_t noct = (uint32_t)p.chunk >> 3, rowlo = 2u * (uint32_t)p.n, rowhi = (uint32_t)p.n;
		const uint32_t lane2 = 2u * (uint32_t)lane, hc4 = 4u * (uint32_t)hcol;
		auto count8 = [&](uint32_t hv, const uint32_t (&v)[4]) {
#pragma unroll
			for (int u = 0; u < 4; u++) {
				/* bin = low byte | 9th bit << 8; the row of the histogram is bin << 7 (lds_count) */
				lds_count(h, lo8(v[u]) | ((hv << (8 - 2 * u)) & 0x100u), hc4, inc);
				lds_count(h, ((v[u] >> 8) & 0xffu) | ((hv << (7 - 2 * u)) & 0x100u), hc4, inc);
			}
		};
		uint32_t o = wv;
#pragma unroll 1
		for (; o + NW < noct; o += 2 * NW) {
			const uint32_t so_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(o * 4u * rowlo));
			const uint32_t so_h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(o * rowhi));
			uint32_t va[4], vb[4];
			const uint32_t ha = __builtin_amdgcn_raw_buffer_load_b8(rhi, (uint32_t)lane, so_h, 0);
			const uint32_t hb = __builtin_amdgcn_raw_buffer_load_b8(rhi, (uint32_t)lane, so_h + (uint32_t)NW * rowhi, 0);
#pragma unroll
			for (int u = 0; u < 4; u++) {
				va[u] = __builtin_amdgcn_raw_buffer_load_b16(rlo, lane2, so_l + (uint32_t)u * rowlo, 0);
				vb[u] = __builtin_amdgcn_raw_buffer_load_b16(rlo, lane2, so_l + (uint32_t)(4 * NW + u) * rowlo, 0);
			}
			count8(ha, va);
			count8(hb, vb);
		}
#pragma unroll 1
		for (; o < noct; o += NW) {
			const uint32_t so_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)(o * 4u * rowlo));
			uint32_t va[4];
			const uint32_t ha = __builtin_amdgcn_raw_buffer_load_b8(rhi, (uint32_t)lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)(o * rowhi)), 0);
#pragma unroll
			for (int u = 0; u < 4; u++)
				va[u] = __builtin_amdgcn_raw_buffer_load_b16(rlo, lane2, so_l + (uint32_t)u * rowlo, 0);
			count8(ha, va);
		}
	} else if (p.bins16) {
		/* scalar descriptor + ONE lane offset + a scalar row offset per load (plain pointers cost a 64-bit per-lane address, two VALU
		 * operations, per row), and two operations per atomic's address (mask / shift, then shift-and-add onto the lane's column offset;
		 * left to it, the compiler shifts first and masks afterwards: three): 3.6 -> 2.1 VALU instructions per atomic */
		const __amdgpu_buffer_rsrc_t rs16 = make_rsrc(p.bins + (size_t)c * (p.chunk >> 1) * p.n + x0);	/* (a chunk's rows: < 4 GiB) */
		const uint32_t nq16 = p.chunk >> 1, rowb = 4u * (uint32_t)p.n, lane4 = 4u * (uint32_t)lane, hc4 = 4u * (uint32_t)hcol;
		uint32_t q = wv;
#pragma unroll 1
		for (; q + NW * (IF - 1) < nq16; q += NW * IF) {
			uint32_t v[IF];
			const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q * rowb));
#pragma unroll
			for (int u = 0; u < IF; u++)
				v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs16, lane4, so + (uint32_t)(NW * u) * rowb, 0);
#pragma unroll
			for (int u = 0; u < IF; u++) {
				lds_count(h, lo16(v[u]), hc4, inc);
				lds_count(h, v[u] >> 16, hc4, inc);
			}
		}
#pragma unroll 1
		for (; q < nq16; q += NW) {
			const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(rs16, lane4, (uint32_t)__builtin_amdgcn_readfirstlane((int)(q * rowb)), 0);
			lds_count(h, lo16(v), hc4, inc);
			lds_count(h, v >> 16, hc4, inc);
		}
	} else {
		const __amdgpu_buffer_rsrc_t rs8 = make_rsrc(p.bins + (size_t)c * (p.chunk >> 2) * p.n + x0);
		const uint32_t nq = p.chunk >> 2, rowb = 4u * (uint32_t)p.n, lane4 = 4u * (uint32_t)lane, hc4 = 4u * (uint32_t)hcol;
		uint32_t q = wv;
#if K2_DBG == 1
		uint32_t dbg_acc = 0;
#endif
#pragma unroll 1
		for (; q + NW * (IF - 1) < nq; q += NW * IF) {	/* independent loads in flight per thread */
			uint32_t v[IF];
			const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q * rowb));
#pragma unroll
			for (int u = 0; u < IF; u++)
				v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs8, lane4, so + (uint32_t)(NW * u) * rowb, 0);
#pragma unroll
			for (int u = 0; u < IF; u++) {
#if K2_DBG == 1		/* probe: the loads and the address arithmetic without the LDS atomics */
				dbg_acc += ((v[u] & 0xff) * 32 + hcol) ^ (((v[u] >> 8) & 0xff) * 32 + hcol) ^ (((v[u] >> 16) & 0xff) * 32 + hcol) ^ ((v[u] >> 24) * 32 + hcol);
#elif K2_DBG == 2	/* probe: plain LDS stores instead of atomics */
				h[((v[u]      ) & 0xff) * 32 + hcol] = inc;
				h[((v[u] >>  8) & 0xff) * 32 + hcol] = inc;
				h[((v[u] >> 16) & 0xff) * 32 + hcol] = inc;
				h[((v[u] >> 24)       ) * 32 + hcol] = inc;
#else
				lds_count(h, lo8(v[u]), hc4, inc);
				lds_count(h, (v[u] >>  8) & 0xff, hc4, inc);
				lds_count(h, (v[u] >> 16) & 0xff, hc4, inc);
				lds_count(h, v[u] >> 24, hc4, inc);
#endif
			}
		}
#if K2_DBG == 1
		if (dbg_acc == 0xdeadbeefu) h[0] = dbg_acc;
#endif
#pragma unroll 1
		for (; q < nq; q += NW) {
			const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(rs8, lane4, (uint32_t)__builtin_amdgcn_readfirstlane((int)(q * rowb)), 0);
			lds_count(h, lo8(v), hc4, inc);
			lds_count(h, (v >>  8) & 0xff, hc4, inc);
			lds_count(h, (v >> 16) & 0xff, hc4, inc);
			lds_count(h, v >> 24, hc4, inc);
		}
	}

	K2_STAMP(2);		/* counting loop (wave 0) */
	/* live sum: sum_t pwr_t (1-a)^(B-1-t) from the tile partials, which hold
	 * sum_{t in tile} pwr_t (1-a)^(t_last - t) (display.cl:149-150) */
	{
		const int tiles = p.chunk / p.tile;
		const float2 *pp = p.partial + (size_t)c * tiles * p.n + x0 + lane;
		float s = 0.0f, m = -1000.0f;
#pragma unroll 2
		for (int j = wv; j < tiles; j += NW) {
			const float2 v = pp[(size_t)j * p.n];
			const int t_last = p.t_offset + t_in + (j + 1) * p.tile - 1;
			/* (1-a)^k as exp2(k log2(1-a)): relative error ~1e-6 where the weight is not negligible */
			s += v.x * __builtin_amdgcn_exp2f(p.log2_w * (float)(p.weight_batch - 1 - t_last));
			m = (m < v.y) ? v.y : m;
		}
		red_s[wv][lane] = s;
		red_m[wv][lane] = m;
	}
	K2_STAMP(3);		/* live-sum partials */
	__syncthreads();
	K2_STAMP(4);		/* barrier: the slowest wave's counting */

	if (tid < 64) {
		float s = 0.0f, m = -1000.0f;
#pragma unroll
		for (int j = 0; j < NW; j++) {				/* fixed order: deterministic floats */
			s += red_s[j][tid];
			m = (m < red_m[j][tid]) ? red_m[j][tid] : m;
		}
		p.chunk_sum[(size_t)c * p.n + x0 + tid] = s;
		p.chunk_max[(size_t)c * p.n + x0 + tid] = m;
	}

	if (p.hc16) {
		/* the LDS image as it is: [bin][32] packed pairs, one contiguous block per work-group
		 * (32 KiB at 256 bins); K3 unpacks */
		uint32_t *d = reinterpret_cast<uint32_t *>(p.hc16) + ((size_t)c * (p.n / 64) + blockIdx.x) * nb * 32;
		if (p.rowmask) {
			/* sparse hand-off: only the bin rows with a count are stored (a wave covers two rows of 32 dwords per
			 * step), one bit per row tells K3 which; with noise-like input 4 rows in 5 are empty */
			for (int i = tid; i < nb * 32; i += 64 * NW) {
				const uint32_t v = h[i];
				const unsigned long long bal = __ballot(v != 0);
				const uint32_t nz = (lane & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
				if (nz) {
					d[i] = v;
					if ((lane & 31) == 0)
						atomicOr(&rowbits[i >> 10], 1u << ((i >> 5) & 31));
				}
			}
			K2_STAMP(5);		/* sparse hand-off */
			__syncthreads();
			if (tid < p.mask_words)
				p.rowmask[((size_t)blockIdx.x * p.mask_words + tid) * p.mask_stride + c] = rowbits[tid];
			K2_STAMP(6);
			return;
		}
		{
			const uint4 *h4 = reinterpret_cast<const uint4 *>(h);
			uint4 *d4 = reinterpret_cast<uint4 *>(d);
#pragma unroll 2
			for (int i = tid; i < nb * 8; i += 64 * NW)
				d4[i] = h4[i];
		}
		return;
	}
	/* (rolled loops: the register budget of this kernel is what lets it share a SIMD with K1) */
	uint32_t *dst = p.hc + (size_t)f * nb * p.n + x0 + lane;
	const int sh = (lane & 32) ? 16 : 0;
	if (cpb == 1) {
#pragma unroll 1
		for (int b = wv; b < nb; b += NW)
			dst[(size_t)b * p.n] = (h[b * 32 + hcol] >> sh) & 0xffffu;
	} else {
#pragma unroll 1
		for (int b = wv; b < nb; b += NW) {
			const uint32_t v = (h[b * 32 + hcol] >> sh) & 0xffffu;
			if (v)
				atomicAdd(&dst[(size_t)b * p.n], v);
		}
	}
}

#if defined(FOSPHOR_AMD_PROBES) && defined(K2_TIMING)
extern "C" int fosphor_amd_debug_k2_timing(unsigned long long *out, int reset)
{
	unsigned long long z[16] = {};
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_time), sizeof(z)) != hipSuccess) return -2;
	if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_k2_time), z, sizeof(z)) != hipSuccess) return -3;
	return 0;
}
#endif

hipError_t launch_k2(const K2Params &p, int n_chunks, hipStream_t s)
{
	const size_t lds = (size_t)p.n_bins * 32 * sizeof(uint32_t);
#ifndef K2_IF8
#define K2_IF8 8		/* ... 8-bit indices, chunks of at most 1024 spectra (A/B builds) */
#endif
#ifndef K2_IF16
#define K2_IF16 4		/* index loads in flight per thread, 16-bit / 9-bit index geometries (A/B builds) */
#endif
	if (p.bins16 || p.bins9 || p.bins8p1)
		hipLaunchKernelGGL((k2_count<16, K2_IF16>), dim3((p.n / 64), n_chunks), dim3(1024), lds, s, p);
	else if (p.chunk > 1024)
		hipLaunchKernelGGL((k2_count<4, 4>), dim3((p.n / 64), n_chunks), dim3(256), lds, s, p);
	else
		hipLaunchKernelGGL((k2_count<4, K2_IF8>), dim3((p.n / 64), n_chunks), dim3(256), lds, s, p);
	return hipGetLastError();
}

/* fixed-order reduction of the chunk partials of each batch */
__global__ __launch_bounds__(256)
void k2b_reduce(const K2bParams p)
{
	const int gid = blockIdx.x * 256 + threadIdx.x;
	if (gid >= p.n_batches * p.n)
		return;
	const int f = gid / p.n, x = gid - f * p.n;
	float s = 0.0f, m = -1000.0f;
	for (int c = 0; c < p.cpb; c++) {
		const size_t i = (size_t)(f * p.cpb + c) * p.n + x;
		s += p.chunk_sum[i];
		m = (m < p.chunk_max[i]) ? p.chunk_max[i] : m;
	}
	p.live_sum[gid] = s;
	p.vmax[gid] = m;
}

hipError_t launch_k2b(const K2bParams &p, hipStream_t s)
{
	const int threads = p.n_batches * p.n;
	hipLaunchKernelGGL(k2b_reduce, dim3((threads + 255) / 256), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* One batch of cpb chunks (the time shard of a display frame): sum the chunks' packed 16-bit
 * count slabs into the 32-bit [bin][x] array the all-reduce and K3 work on, and reduce the float
 * partials in the same fixed order as k2b_reduce.  Integer sums: exact, order-independent. */
__global__ __launch_bounds__(256)
void k2c_sum(const K2bParams p)
{
	const int pairs = p.n_bins * p.n / 2;		/* dwords per chunk slab set: columns c, c + 32 packed */
	const int cells = pairs;			/* thread index space: one thread per packed pair */
	const int gid = blockIdx.x * 256 + threadIdx.x;
	const int f = blockIdx.y;			/* batch of the launch */
	if (gid < pairs) {
		const int nb = p.n_bins;
		const int slab = gid / (nb * 32);
		const int rem = gid - slab * nb * 32;
		const int bin = rem >> 5, hcol = rem & 31;
		const uint32_t *src = reinterpret_cast<const uint32_t *>(p.hc16) + (size_t)f * p.cpb * pairs + gid;
		uint32_t lo = 0, hi = 0;
		int c = 0;
		for (; c + 8 <= p.cpb; c += 8) {
			uint32_t v[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
				v[u] = __builtin_nontemporal_load(&src[(size_t)(c + u) * pairs]);
#pragma unroll
			for (int u = 0; u < 8; u++) {
				lo += v[u] & 0xffffu;
				hi += v[u] >> 16;
			}
		}
		for (; c < p.cpb; c++) {
			const uint32_t v = src[(size_t)c * pairs];
			lo += v & 0xffffu;
			hi += v >> 16;
		}
		uint32_t *dst = p.hc + (size_t)f * p.n_bins * p.n + bin * p.n + slab * 64 + hcol;
		dst[0]  = lo;
		dst[32] = hi;
	} else if (gid < cells + p.n) {
		const int x = gid - cells;
		const float *cs = p.chunk_sum + (size_t)f * p.cpb * p.n, *cm = p.chunk_max + (size_t)f * p.cpb * p.n;
		float s = 0.0f, m = -1000.0f;
		int c = 0;
		for (; c + 8 <= p.cpb; c += 8) {
			float a[8], b[8];
#pragma unroll
			for (int u = 0; u < 8; u++) {
				a[u] = cs[(size_t)(c + u) * p.n + x];
				b[u] = cm[(size_t)(c + u) * p.n + x];
			}
#pragma unroll
			for (int u = 0; u < 8; u++) {		/* same order as k2b_reduce */
				s += a[u];
				m = (m < b[u]) ? b[u] : m;
			}
		}
		for (; c < p.cpb; c++) {
			s += cs[(size_t)c * p.n + x];
			m = (m < cm[(size_t)c * p.n + x]) ? cm[(size_t)c * p.n + x] : m;
		}
		p.live_sum[(size_t)f * p.n + x] = s;
		p.vmax[(size_t)f * p.n + x] = m;
	}
}

hipError_t launch_k2c(const K2bParams &p, hipStream_t s)
{
	const int threads = p.n_bins * p.n / 2 + p.n;
	hipLaunchKernelGGL(k2c_sum, dim3((threads + 255) / 256, p.n_batches), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* K3: state update                                                          */
/* ------------------------------------------------------------------------ */

/* MODE 0: 16-bit slab-major counts + LDS (d, e) table (batch <= 1024); 3: the same counts, table in memory
 * (batches of up to 8192 spectra counted as one chunk); 1: 32-bit counts + (d, e) table in memory;
 * 2: 32-bit counts, (d, e) evaluated per cell (batches beyond the table).  Separate instantiations keep
 * the common one (0) at a register budget that lets it share a SIMD with K1. */
#ifndef K3_ROWS
#define K3_ROWS 2		/* rows in flight per wave of the sparse form: measured at N = 65536, 1 / 2 / 3 / 4 / 8 / 16 -> scan + merge 56 / 49 / 49 /
				 * 53 / 61 / 114 us per frame (42 / 58 / 74 / ... / 256 VGPRs: more resident waves beat more requests per wave) */
#endif
#ifndef K3_BATCHES
#define K3_BATCHES 2
#endif
#ifndef K3_NO_PIPE
#define K3_NO_PIPE 0		/* A/B builds: 1 = the sparse form without its software pipeline */
#endif
template <int MODE, bool SPARSE = false>
__global__ __launch_bounds__(256)
void k3_merge(const K3Params p)
{
	const int cells = p.n_bins * p.n;
	const float fbatch = (float)p.batch;

	/* the (d, e) table of the 16-bit path sits in LDS: loaded once per work-group, which then strides
	 * over the cells (the grid is capped, so a 128 MiB state does not reload it 131 072 times) */
	/* (long batches, MODE 3: in LDS as well up to 4096 spectra -- a look-up in memory is one more dependent round trip per batch
	 * and cell; longer batches read it from memory) */
	constexpr int kRiseLds = (MODE == 0) ? 1025 : (MODE == 3 && !SPARSE) ? 4097 : 1;
	__shared__ float2 rise_lds[kRiseLds];
	const bool rise_in_lds = (MODE == 0) || (MODE == 3 && !SPARSE && p.batch < kRiseLds);
	if (rise_in_lds) {
		for (int i = threadIdx.x; i <= p.batch && i < kRiseLds; i += 256)
			rise_lds[i] = p.rise[i];
		__syncthreads();
	}

	/* one column: live EMA (display.cl:186-214) and max-hold (display.cl:257-310) */
	auto update_column = [&](const int x) {
		const int half = p.n >> 1;
		const int i = x ^ half;
		const float decay = p.live_decay;
		float live = p.spectrum[i].y;
		float mh   = p.spectrum[p.n + i].y;
		for (int f = 0; f < p.n_batches; f++) {
			const float sum = p.live_sum[(size_t)f * p.n + x];
			const float mx  = p.vmax[(size_t)f * p.n + x];
			if (!__builtin_isfinite(live))
				live = sum / 16.0f;			/* display.cl:206-207 */
			live = live * decay + sum * p.alpha;		/* display.cl:210-211 */
			if (!__builtin_isfinite(mh))
				mh = -3.402823466e+38f;			/* display.cl:290-291 */
			mh = mh * 0.999f + 0.001f * live;		/* display.cl:303 */
			mh = (mh < mx) ? mx : mh;			/* display.cl:304-305 */
		}
		const float vx = ((float)i / (float)half) - 1.0f;	/* display.cl:209,293 */
		p.spectrum[i]      = make_float2(vx, live);
		p.spectrum[p.n + i] = make_float2(vx, mh);
	};

	if (SPARSE) {
		/* Sparse form: k3_scan has listed the rows that are alive (hot, or with a count in some batch of the launch);
		 * a wave takes every n_waves-th entry of the list, its 64 lanes are the row's 64 cells.  A cold, empty row
		 * costs one byte of flag and a few mask bits in the scan and nothing here. */
		const int lane = threadIdx.x & 63;
		const int nb = p.n_bins;
		const int n_waves = gridDim.x * 4;
		const int count = (int)p.rowlist[p.rowlist_cnt];
		const int col = (lane >> 1) + ((lane & 1) << 5);
		constexpr int R = K3_ROWS;		/* rows in flight per wave: every step below is R independent requests */
		constexpr int U = K3_BATCHES;		/* batches of counts in flight per row */
		if (p.n_batches <= U && !K3_NO_PIPE) {
			/* One or two batches per launch (a display frame of the 65536-point configuration is ONE): a row is a list entry, then
			 * the histogram value and the counts it points to -- two dependent round trips for a few instructions of arithmetic, and a
			 * wave walks ~20 rows.  Software pipeline, three deep: while row set i is computed and stored, the values of set i + 1 are
			 * on their way and so are the list entries of set i + 2 (loads return in order: each wait leaves the younger requests
			 * outstanding). */
			const int step = R * n_waves;
			const int fe = p.n_batches;
			int idx = (blockIdx.x * 256 + threadIdx.x) >> 6;
			uint32_t e_c[R], e_n[R], hc_c[R][U], hc_n[R][U];
			int hidx_c[R], hidx_n[R];
			float hv0_c[R], hv0_n[R];
			auto entries = [&](int at, uint32_t (&e)[R]) {
#pragma unroll
				for (int r = 0; r < R; r++)
					e[r] = (at + r * n_waves < count) ? p.rowlist[1 + at + r * n_waves] : 0xffffffffu;	/* (no such entry: row index 0xfffff with every flag) */
			};
			auto values = [&](const uint32_t (&e)[R], int (&hidx)[R], float (&hv0)[R], uint32_t (&hc)[R][U]) {
#pragma unroll
				for (int r = 0; r < R; r++) {
					const bool ok = e[r] != 0xffffffffu;
					const int row = (int)(e[r] & 0xfffffu);
					const int slab = row / nb, bin = row - slab * nb;
					hidx[r] = bin * p.n + slab * 64 + col;
					hv0[r] = ok ? p.hist[hidx[r]] : 0.0f;
#pragma unroll
					for (int u = 0; u < U; u++)
						hc[r][u] = (ok && u < fe && ((e[r] >> (20 + u)) & 1u))
						        ? (uint32_t)__builtin_nontemporal_load(&p.hc16[(size_t)(PROBE_SAME(p) ? 0 : u) * cells + row * 64 + lane]) : 0u;
				}
			};
			entries(idx, e_c);
			values(e_c, hidx_c, hv0_c, hc_c);
			entries(idx + step, e_n);
			for (; idx < count; idx += step) {
				uint32_t e_nn[R];
				values(e_n, hidx_n, hv0_n, hc_n);
				entries(idx + 2 * step, e_nn);
#pragma unroll
				for (int r = 0; r < R; r++) {
					if (e_c[r] == 0xffffffffu)		/* uniform */
						continue;
					float hv = hv0_c[r];
#pragma unroll
					for (int u = 0; u < U; u++) {
						if (u < fe && !((hv <= 0.01f) && (hc_c[r][u] == 0))) {	/* display.cl:237-238 */
							const float2 de = (MODE == 0) ? rise_lds[hc_c[r][u]] : p.rise[hc_c[r][u]];
							hv = (hv - de.x) * de.y + de.x;			/* display.cl:247 */
							hv = (hv < 0.0f) ? 0.0f : hv;			/* clamp, display.cl:250 */
							hv = (1.0f < hv) ? 1.0f : hv;
						}
					}
					if (__float_as_uint(hv) != __float_as_uint(hv0_c[r]))
						p.hist[hidx_c[r]] = hv;		/* cold cells (display.cl:237-238) keep their line clean */
					const bool was_hot = (e_c[r] >> 31) != 0;
					const bool now_hot = __ballot(!(hv <= 0.01f)) != 0;
					if (lane == 0 && (p.hot_all || now_hot != was_hot))
						p.hot[e_c[r] & 0xfffffu] = now_hot ? 1 : 0;
				}
#pragma unroll
				for (int r = 0; r < R; r++) {
					e_c[r] = e_n[r]; hidx_c[r] = hidx_n[r]; hv0_c[r] = hv0_n[r];
#pragma unroll
					for (int u = 0; u < U; u++)
						hc_c[r][u] = hc_n[r][u];
					e_n[r] = e_nn[r];
				}
			}
		} else
		for (int idx0 = (blockIdx.x * 256 + threadIdx.x) >> 6; idx0 < count; idx0 += R * n_waves) {
			uint32_t e[R];
			int slab[R], bin[R], hidx[R], gid[R];
			float hv0[R], hv[R];
			bool valid[R];
#pragma unroll
			for (int r = 0; r < R; r++) {
				valid[r] = idx0 + r * n_waves < count;		/* uniform */
				e[r] = valid[r] ? p.rowlist[1 + idx0 + r * n_waves] : 0u;
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
				const int row = (int)(e[r] & 0xfffffu);
				slab[r] = row / nb; bin[r] = row - slab[r] * nb;
				gid[r] = row * 64 + lane;
				hidx[r] = bin[r] * p.n + slab[r] * 64 + col;
				hv0[r] = valid[r] ? p.hist[hidx[r]] : 0.0f;
				hv[r] = hv0[r];
			}
			for (int f0 = 0; f0 < p.n_batches; f0 += 64) {
				unsigned long long m[R];
				const int fl = f0 + lane;
				if (p.n_batches <= 11) {
#pragma unroll
					for (int r = 0; r < R; r++)
						m[r] = (e[r] >> 20) & 0x7ffu;		/* carried by the list entry */
				} else {
#pragma unroll
					for (int r = 0; r < R; r++) {
						const uint32_t wd = (valid[r] && fl < p.n_batches)
						        ? p.rowmask[((size_t)slab[r] * p.mask_words + (bin[r] >> 5)) * p.mask_stride + (PROBE_SAME(p) ? 0 : fl)] : 0u;
						m[r] = __ballot((wd >> (bin[r] & 31)) & 1u);
					}
				}
				const int fe = (p.n_batches - f0 < 64) ? p.n_batches : f0 + 64;
				for (int f = f0; f < fe; f += U) {
					uint32_t hc[R][U];
#pragma unroll
					for (int r = 0; r < R; r++)
#pragma unroll
						for (int u = 0; u < U; u++)
							hc[r][u] = (f + u < fe && ((m[r] >> (f + u - f0)) & 1ull))
							        ? (uint32_t)__builtin_nontemporal_load(&p.hc16[(size_t)(PROBE_SAME(p) ? 0 : f + u) * cells + gid[r]]) : 0u;
#pragma unroll
					for (int r = 0; r < R; r++)
#pragma unroll
						for (int u = 0; u < U; u++) {
							if (f + u < fe && !((hv[r] <= 0.01f) && (hc[r][u] == 0))) {	/* display.cl:237-238 */
								const float2 de = (MODE == 0) ? rise_lds[hc[r][u]] : p.rise[hc[r][u]];
								hv[r] = (hv[r] - de.x) * de.y + de.x;		/* display.cl:247 */
								hv[r] = (hv[r] < 0.0f) ? 0.0f : hv[r];		/* clamp, display.cl:250 */
								hv[r] = (1.0f < hv[r]) ? 1.0f : hv[r];
							}
						}
				}
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
				if (!valid[r])
					continue;
				if (__float_as_uint(hv[r]) != __float_as_uint(hv0[r]))
					p.hist[hidx[r]] = hv[r];	/* cold cells (display.cl:237-238) keep their line clean */
				const bool was_hot = (e[r] >> 31) != 0;
				const bool now_hot = __ballot(!(hv[r] <= 0.01f)) != 0;
				if (lane == 0 && (p.hot_all || now_hot != was_hot))
					p.hot[e[r] & 0xfffffu] = now_hot ? 1 : 0;
			}
		}
		for (int x = blockIdx.x * 256 + threadIdx.x; x < p.n; x += gridDim.x * 256)
			update_column(x);
		return;
	}

	if (MODE == 3 && !SPARSE && p.n_batches <= 4) {
		/* Long batches come a few per launch (4 at N = 8192): too few for the batches-in-flight scheme below to hide anything, and a
		 * thread that walks its cells one after the other pays a memory round trip per cell.  FOUR CELLS IN FLIGHT per thread instead:
		 * their histogram values and all their counts are requested together. */
		constexpr int R = 4;
		const int stride = gridDim.x * 256;
		const int nb = p.n_bins, fe = p.n_batches;
		const int pairs = cells >> 1;		/* a dword of the slabs = columns c and c + 32 of one (slab, bin) */
		const uint32_t *hc32 = reinterpret_cast<const uint32_t *>(p.hc16);
		for (int base = blockIdx.x * 256 + threadIdx.x; base < pairs; base += R * stride) {
			int hidx[R]; float hv0[R][2]; uint32_t hc[R][4];
#pragma unroll
			for (int r = 0; r < R; r++) {
				const int g = base + r * stride;
				const bool ok = g < pairs;
				const int gg = ok ? g : base;
				const int slab = gg / (nb * 32);
				const int rem = gg - slab * nb * 32;
				hidx[r] = ok ? (rem >> 5) * p.n + slab * 64 + (rem & 31) : -1;
				hv0[r][0] = p.hist[ok ? hidx[r] : 0];
				hv0[r][1] = p.hist[ok ? hidx[r] + 32 : 0];
#pragma unroll
				for (int f = 0; f < 4; f++)
					hc[r][f] = (f < fe) ? __builtin_nontemporal_load(&hc32[(size_t)(PROBE_SAME(p) ? 0 : f) * pairs + gg]) : 0u;
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
#pragma unroll
				for (int h = 0; h < 2; h++) {
					float hv = hv0[r][h];
#pragma unroll
					for (int f = 0; f < 4; f++) {
						const uint32_t c16 = h ? (hc[r][f] >> 16) : (hc[r][f] & 0xffffu);
						if (f < fe && !((hv <= 0.01f) && (c16 == 0))) {	/* display.cl:237-238 */
							const float2 de = rise_in_lds ? rise_lds[c16] : p.rise[c16];
							hv = (hv - de.x) * de.y + de.x;			/* display.cl:247 */
							hv = (hv < 0.0f) ? 0.0f : hv;			/* clamp, display.cl:250 */
							hv = (1.0f < hv) ? 1.0f : hv;
						}
					}
					if (hidx[r] >= 0 && __float_as_uint(hv) != __float_as_uint(hv0[r][h]))
						p.hist[hidx[r] + 32 * h] = hv;	/* cold cells (display.cl:237-238) keep their line clean */
				}
			}
		}
		for (int x = blockIdx.x * 256 + threadIdx.x; x < p.n; x += stride)
			update_column(x);
		return;
	}

	for (int gid = blockIdx.x * 256 + threadIdx.x; gid < cells + p.n; gid += gridDim.x * 256) {
	if (MODE == 0 || MODE == 3) {
		/* 16-bit slab-major counts as K2 leaves them ([slab of 64 columns][bin][32] dwords, columns
		 * c and c + 32 in the low / high half): thread gid reads the gid-th 16-bit word of a batch
		 * (2 B per batch, 8 batches in flight); the (d, e) table sits in LDS (one dependent lookup
		 * per batch per cell). */
		if (gid < cells) {
			const int nb = p.n_bins;
			const int slab = gid / (nb * 64);
			const int rem = gid - slab * nb * 64;
			const int bin = rem >> 6;
			const int col = ((rem & 63) >> 1) + ((rem & 1) << 5);
			const int hidx = bin * p.n + slab * 64 + col;
			const float hv0 = p.hist[hidx];
			float hv = hv0;
			{
				const int fe = p.n_batches;
				int f = 0;
				for (; f + 8 <= fe; f += 8) {
					uint32_t hc[8];
#pragma unroll
					for (int u = 0; u < 8; u++)
						hc[u] = (uint32_t)__builtin_nontemporal_load(&p.hc16[(size_t)(PROBE_SAME(p) ? 0 : f + u) * cells + gid]);
#pragma unroll
					for (int u = 0; u < 8; u++) {
						if (!((hv <= 0.01f) && (hc[u] == 0))) {	/* display.cl:237-238 */
							const float2 de = rise_in_lds ? rise_lds[hc[u]] : p.rise[hc[u]];
							hv = (hv - de.x) * de.y + de.x;		/* display.cl:247 */
							hv = (hv < 0.0f) ? 0.0f : hv;		/* clamp, display.cl:250 */
							hv = (1.0f < hv) ? 1.0f : hv;
						}
					}
				}
				/* (launches of fewer than 8 batches -- 4 at N = 8192 -- : the counts of what is left requested together as well) */
				if (f + 4 <= fe) {
					uint32_t hc[4];
#pragma unroll
					for (int u = 0; u < 4; u++)
						hc[u] = (uint32_t)__builtin_nontemporal_load(&p.hc16[(size_t)(PROBE_SAME(p) ? 0 : f + u) * cells + gid]);
#pragma unroll
					for (int u = 0; u < 4; u++) {
						if (!((hv <= 0.01f) && (hc[u] == 0))) {
							const float2 de = rise_in_lds ? rise_lds[hc[u]] : p.rise[hc[u]];
							hv = (hv - de.x) * de.y + de.x;
							hv = (hv < 0.0f) ? 0.0f : hv;
							hv = (1.0f < hv) ? 1.0f : hv;
						}
					}
					f += 4;
				}
				for (; f < fe; f++) {
					const uint32_t hc = (uint32_t)p.hc16[(size_t)(PROBE_SAME(p) ? 0 : f) * cells + gid];
					if (!((hv <= 0.01f) && (hc == 0))) {
						const float2 de = rise_in_lds ? rise_lds[hc] : p.rise[hc];
						hv = (hv - de.x) * de.y + de.x;
						hv = (hv < 0.0f) ? 0.0f : hv;
						hv = (1.0f < hv) ? 1.0f : hv;
					}
				}
			}
			if (__float_as_uint(hv) != __float_as_uint(hv0))
				p.hist[hidx] = hv;	/* cold cells (display.cl:237-238) keep their line clean */
		}
	}
	if (MODE == 0 || MODE == 3) {
		/* cells handled above */
	} else if (gid < cells && (p.cell_end == 0 || (gid >= p.cell_begin && gid < p.cell_end))) {
		/* one (bin, x) cell; batches applied in order (display.cl:217-254).
		 * d and e of display.cl:241-245 depend only on the hit count: with a table
		 * rise[hc] = (d, e) (host-computed with the same powf the oracle uses) the update
		 * is a lookup and display.cl:247,250. */
		float hv = p.hist[gid];
		if (MODE == 1) {
			/* 8 batches of counts in flight per thread: the loop is otherwise one dependent
			 * HBM/L2 round trip per batch */
			int f = 0;
			for (; f + 8 <= p.n_batches; f += 8) {
				uint32_t hc[8];
#pragma unroll
				for (int u = 0; u < 8; u++)
					hc[u] = __builtin_nontemporal_load(&p.hc[(size_t)(f + u) * cells + gid]);
#pragma unroll
				for (int u = 0; u < 8; u++) {
					if (!((hv <= 0.01f) && (hc[u] == 0))) {	/* display.cl:237-238 */
						const float2 de = p.rise[hc[u]];
						hv = (hv - de.x) * de.y + de.x;		/* display.cl:247 */
						hv = (hv < 0.0f) ? 0.0f : hv;		/* clamp, display.cl:250 */
						hv = (1.0f < hv) ? 1.0f : hv;
					}
				}
			}
			for (; f < p.n_batches; f++) {
				const uint32_t hc = p.hc[(size_t)f * cells + gid];
				if (!((hv <= 0.01f) && (hc == 0))) {
					const float2 de = p.rise[hc];
					hv = (hv - de.x) * de.y + de.x;
					hv = (hv < 0.0f) ? 0.0f : hv;
					hv = (1.0f < hv) ? 1.0f : hv;
				}
			}
		} else {
			const float rt0r = 1.0f / p.t0r, rt0d = 1.0f / p.t0d;
			for (int f = 0; f < p.n_batches; f++) {
				const uint32_t hc = p.hc[(size_t)f * cells + gid];
				if ((hv <= 0.01f) && (hc == 0))			/* display.cl:237-238 */
					continue;
				const float a = (float)hc / fbatch;		/* display.cl:241-245 */
				const float b = a * rt0r;
				const float c = b + rt0d;
				const float d = b * (1.0f / c);
				const float e = powf(1.0f - c, fbatch);
				hv = (hv - d) * e + d;				/* display.cl:247 */
				hv = (hv < 0.0f) ? 0.0f : hv;			/* clamp, display.cl:250 */
				hv = (1.0f < hv) ? 1.0f : hv;
			}
		}
		p.hist[gid] = hv;
	}
	if (gid >= cells && gid < cells + p.n)
		update_column(gid - cells);
	}	/* cell loop */
}

/* Sparse form, first step: the list of live rows.  rowlist[rowlist_cnt] = count -- two counters, word 0 and the word behind the list,
 * used by alternate launches: a scan zeroes the one the NEXT scan will add to (its last reader, the merge kernel before this scan, has
 * finished: same stream), so the host queues no memset per frame --, rowlist[1 + i] = row
 * index (20 bits), bits 20-30 = which batches have counts in the row (launches of <= 11 batches), bit 31 = the row's hot
 * flag as stored. */
__global__ __launch_bounds__(1024)
void k3_scan(const K3Params p)
{
	constexpr int RPT = 4;				/* rows per thread: 4096 rows and ONE atomic on the shared counter per block */
	__shared__ uint32_t wave_cnt[16], wave_base[16];
	const int rows = p.n_bins * (p.n >> 6);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int row0 = blockIdx.x * (1024 * RPT) + threadIdx.x;
	/* (list order = row order, slab-major like the count kernel's output.  Bin-major -- consecutive entries the same bin of adjacent
	 * slabs, i.e. adjacent 256-byte pieces of the histogram but counts 64 KiB apart -- measured 45 against 40 us at N = 65536.) */
	bool hot[RPT], act[RPT];
	uint32_t bits[RPT];			/* bit f: batch f of the launch has counts in this row (launches of <= 11 batches) */
	uint32_t mine = 0;
	const bool carry = p.n_batches <= 11;	/* ... then the list entry carries them and k3_merge needs no mask request */
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int row = row0 + 1024 * k;
		hot[k] = (row < rows) && p.hot[row] != 0;
		act[k] = (row < rows) && (hot[k] || p.hot_all);
		bits[k] = 0;
	}
#pragma unroll
	for (int k = 0; k < RPT; k++) {
		const int row = row0 + 1024 * k;
		if (row < rows && (carry || !act[k])) {
			const int slab = row / p.n_bins, bin = row - slab * p.n_bins;
			const uint32_t *mw = p.rowmask + ((size_t)slab * p.mask_words + (bin >> 5)) * p.mask_stride;
			uint32_t any = 0;
			for (int f = 0; f < p.n_batches; f++) {
				const uint32_t b = (mw[PROBE_SAME(p) ? 0 : f] >> (bin & 31)) & 1u;
				any |= b;
				if (carry)
					bits[k] |= b << f;
			}
			act[k] = act[k] || any;
		}
		mine += act[k] ? 1u : 0u;
	}
	/* wave prefix of `mine`, then the waves' totals through LDS */
	uint32_t incl = mine;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t o = __shfl_up(incl, d, 64);
		if (lane >= d) incl += o;
	}
	if (lane == 63)
		wave_cnt[wv] = incl;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t tot = 0;
		for (int w = 0; w < 16; w++) { wave_base[w] = tot; tot += wave_cnt[w]; }
		const uint32_t base = tot ? atomicAdd(&p.rowlist[p.rowlist_cnt], tot) : 0u;
		if (blockIdx.x == 0)
			p.rowlist[p.rowlist_cnt ? 0 : 1 + rows] = 0;
		for (int w = 0; w < 16; w++) wave_base[w] += base;
	}
	__syncthreads();
	uint32_t pos = wave_base[wv] + incl - mine;
#pragma unroll
	for (int k = 0; k < RPT; k++)
		if (act[k])
			p.rowlist[1 + pos++] = (uint32_t)(row0 + 1024 * k) | (bits[k] << 20) | (hot[k] ? 0x80000000u : 0u);
}

hipError_t launch_k3(const K3Params &p, hipStream_t s)
{
	const int threads = p.n_bins * p.n + p.n;
	int blocks = (threads + 255) / 256;
	if (blocks > 8192) blocks = 8192;
	if (p.hc16 && p.rowmask && p.n_bins * (p.n / 64) <= (1 << 20)) {	/* (list entries hold 20 bits of row index: every geometry the library accepts) */
		/* sparse form: list the live rows, then one wave per listed row (strided) */
		const int rows = p.n_bins * (p.n / 64);
		hipLaunchKernelGGL(k3_scan, dim3((rows + 4095) / 4096), dim3(1024), 0, s, p);
		int sb = (rows + 4 * K3_ROWS - 1) / (4 * K3_ROWS);	/* 4 waves x K3_ROWS rows in flight per block; the list is usually far shorter */
		if (sb > 2048) sb = 2048;
		if (p.batch <= 1024)
			hipLaunchKernelGGL((k3_merge<0, true>), dim3(sb), dim3(256), 0, s, p);
		else
			hipLaunchKernelGGL((k3_merge<3, true>), dim3(sb), dim3(256), 0, s, p);
		return hipGetLastError();
	}
	if (p.hc16 && p.batch <= 1024)
		hipLaunchKernelGGL(k3_merge<0>, dim3(blocks), dim3(256), 0, s, p);
	else if (p.hc16)	/* (each work-group loads the 32 KiB table: four per CU stride over the cells) */
		hipLaunchKernelGGL(k3_merge<3>, dim3(p.batch <= 4096 && blocks > 1024 ? 1024 : blocks), dim3(256), 0, s, p);
	else if (p.rise)
		hipLaunchKernelGGL(k3_merge<1>, dim3(blocks), dim3(256), 0, s, p);
	else
		hipLaunchKernelGGL(k3_merge<2>, dim3(blocks), dim3(256), 0, s, p);
	return hipGetLastError();
}

/* ------------------------------------------------------------------------ */

__global__ void k_fill(float *dst, float value, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = value;
}

/* uint32 [bin][x] view of one batch of 16-bit slab-major counts (the layout K3 mode 0 / 3 reads; rows K2 did not
 * store -- clear bit in its row mask -- are zero) */
__global__ __launch_bounds__(256)
void k_export_hc16(const uint16_t *__restrict__ hc16, const uint32_t *__restrict__ rowmask, int mask_words, int mask_stride,
                   uint32_t *__restrict__ out, int n_bins, int n)
{
	const size_t cells = (size_t)n_bins * n;
	for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < cells; gid += (size_t)gridDim.x * 256) {
		const int slab = (int)(gid / ((size_t)n_bins * 64));
		const int rem = (int)(gid - (size_t)slab * n_bins * 64);
		const int bin = rem >> 6;
		const int col = ((rem & 63) >> 1) + ((rem & 1) << 5);
		const bool stored = !rowmask || ((rowmask[((size_t)slab * mask_words + (bin >> 5)) * mask_stride] >> (bin & 31)) & 1u);
		out[(size_t)bin * n + slab * 64 + col] = stored ? hc16[gid] : 0u;
	}
}

hipError_t launch_export_hc16(const uint16_t *hc16, const uint32_t *rowmask, int mask_words, int mask_stride, uint32_t *out, int n_bins, int n, hipStream_t s)
{
	size_t blocks = ((size_t)n_bins * n + 255) / 256;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(k_export_hc16, dim3((unsigned)blocks), dim3(256), 0, s, hc16, rowmask, mask_words, mask_stride, out, n_bins, n);
	return hipGetLastError();
}

hipError_t launch_fill(float *dst, float value, size_t n, hipStream_t s)
{
	size_t blocks = (n + 255) / 256;
	if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, dst, value, n);
	return hipGetLastError();
}

} // namespace fosphor_amd
