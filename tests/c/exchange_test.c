/*
 * exchange_test.c -- what a C/C++ sink that links libfosphor_amd.so does for a multi-GPU frame, with no Python and no
 * PyTorch in the process: communicator from the library (RCCL bound at run time), then per frame
 *     fosphor_amd_accumulate_device -> fosphor_amd_exchange -> fosphor_amd_merge
 * on one rank (world size 1: the 8-GPU run is the driver's), checked against the single-launch path of the same C ABI
 * (fosphor_amd_process_device with one batch of the whole frame): hit counts bit-identical, histogram identical.
 *
 *   gcc -O2 -I include tests/c/exchange_test.c -o build/exchange_test -L gr-fosphor_amd -lfosphor_amd \
 *       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gr-fosphor_amd -Wl,-rpath,/opt/rocm/lib -lm
 * (built and run by tests/test_gpu_dist.py::test_c_program_native_exchange)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fosphor_amd.h"

/* the three HIP runtime calls a C host needs (hip_runtime_api.h is C++-flavoured: declare them) */
extern int hipMalloc(void **ptr, size_t size);
extern int hipMemcpy(void *dst, const void *src, size_t size, int kind);
extern int hipFree(void *ptr);
#define H2D 1

#define N 1024
#define FRAME 2048		/* spectra per frame */
#define FRAMES 3

static uint32_t lcg = 12345u;
static float gauss(void)
{
	float s = 0.0f;			/* sum of 12 uniforms: good enough for a test signal */
	for (int i = 0; i < 12; i++) { lcg = lcg * 1664525u + 1013904223u; s += (float)(lcg >> 8) * (1.0f / 16777216.0f); }
	return (s - 6.0f) * 0.05f;
}

int main(void)
{
	const size_t samples = (size_t)FRAME * N;
	float *h = malloc(samples * 2 * sizeof(float));
	void *d = NULL, *comm = NULL;
	char id[128];
	struct fosphor_amd_config cfg;
	struct fosphor *a, *b;
	const size_t cells = 128 * N;
	uint32_t *hc_a = malloc(cells * 4), *hc_b = malloc(cells * 4);
	float *hi_a = malloc(cells * 4), *hi_b = malloc(cells * 4);
	int rv = 0;

	memset(&cfg, 0, sizeof(cfg));
	cfg.device = -1; cfg.max_spectra = FRAME;
	a = fosphor_amd_init(&cfg);
	b = fosphor_amd_init(&cfg);
	if (!a || !b || hipMalloc(&d, samples * 2 * sizeof(float))) { fprintf(stderr, "init failed\n"); return 2; }
	if (fosphor_amd_comm_unique_id(id) || fosphor_amd_comm_init(&comm, 1, 0, id)) { fprintf(stderr, "no RCCL communicator\n"); return 3; }

	for (int f = 0; f < FRAMES; f++) {
		for (size_t i = 0; i < samples; i++) {
			const double ph = 2.0 * M_PI * 0.0625 * (f + 1) * (double)i;
			h[2 * i] = gauss() + 0.1f * (float)cos(ph);
			h[2 * i + 1] = gauss() + 0.1f * (float)sin(ph);
		}
		if (fosphor_amd_finish(a) < 0 || fosphor_amd_finish(b) < 0) return 4;	/* d is rewritten: everything read it */
		if (hipMemcpy(d, h, samples * 2 * sizeof(float), H2D)) return 4;
		/* sharded form, one rank holding the whole frame */
		rv |= fosphor_amd_accumulate_device(a, d, FRAME, 0, FRAME);
		rv |= fosphor_amd_exchange(a, comm);
		rv |= fosphor_amd_merge(a, FRAME);
		/* single launch with fft_batch = FRAME */
		rv |= fosphor_amd_process_device(b, d, 1, FRAME);
		if (rv) { fprintf(stderr, "frame %d: rv %d\n", f, rv); return 5; }
	}
	if (fosphor_amd_read(a, 3, hc_a, cells * 4) || fosphor_amd_read(b, 3, hc_b, cells * 4) ||
	    fosphor_amd_read(a, 1, hi_a, cells * 4) || fosphor_amd_read(b, 1, hi_b, cells * 4)) return 6;
	size_t bad_hc = 0, bad_hi = 0, sum = 0;
	for (size_t i = 0; i < cells; i++) {
		bad_hc += hc_a[i] != hc_b[i];
		bad_hi += fabsf(hi_a[i] - hi_b[i]) > 2e-6f;
		sum += hc_a[i];
	}
	printf("frames %d: %zu hit-count cells differ, %zu histogram cells differ, counts sum %zu (want %zu)\n",
	       FRAMES, bad_hc, bad_hi, sum, (size_t)FRAME * N);
	fosphor_amd_comm_destroy(comm);
	fosphor_release(a); fosphor_release(b);
	hipFree(d);
	if (bad_hc || bad_hi || sum != (size_t)FRAME * N) return 1;
	printf("c exchange ok\n");
	return 0;
}
