#!/usr/bin/env python3
"""Probe (round 5): can a memory-bound kernel with a small footprint run BESIDE k1h_fused on the CUs it occupies, and what does it cost?
A probe build without count / merge (FOSPHOR_AMD_DBG_SKIP=2) runs 1024-spectrum frames back to back while a side stream copies
256 MiB buffers with torch's elementwise kernel (256 threads, few registers, no LDS).

    tools/r04_ceiling_build.sh   (lib_probes.so)
    gpurun -- 'FOSPHOR_AMD_LIB=$PWD/build/ab/lib_probes.so FOSPHOR_AMD_DBG_SKIP=2 python3 tools/c5_coresidency_probe.py'
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

total, n = 1024, 65536
f = gr_fosphor_amd.Fosphor(fft_len_log=16, n_bins=512, max_spectra=total, max_batches=1, iq_fp16=True)
f.set_input_ordering(False)
iq = torch.empty((2 * total * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05).to(torch.float16)
a = torch.empty(64 << 20, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
side = torch.cuda.Stream()


def frames(k):
    for i in range(k):
        assert f.process_device(iq[(i & 1) * total * n:((i & 1) + 1) * total * n], 1, total) == 0


def run(n_frames, n_copies):
    torch.cuda.synchronize(); f.finish()
    t0 = time.perf_counter()
    if n_copies:
        with torch.cuda.stream(side):
            for _ in range(n_copies):
                b.copy_(a, non_blocking=True)
    if n_frames:
        frames(n_frames)
    f.finish(); torch.cuda.synchronize()
    return time.perf_counter() - t0


frames(50); run(10, 10)
t_f = run(400, 0)
t_c = run(0, 400)
print("alone:    %d frames %.1f ms (%.1f us per frame);  %d copies of 256 MiB %.1f ms (%.1f us each, %.2f TB/s moved)"
      % (400, t_f * 1e3, t_f / 400 * 1e6, 400, t_c * 1e3, t_c / 400 * 1e6, 2 * 256 * 2**20 / (t_c / 400) / 1e12))
for nc in (100, 200, 400):
    t_b = run(400, nc)
    print("together: 400 frames + %d copies: %.1f ms  (sum of the two alone: %.1f ms; frames alone %.1f)" % (nc, t_b * 1e3, (t_f + t_c * nc / 400) * 1e3, t_f * 1e3))
f.close()
