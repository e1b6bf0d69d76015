#!/usr/bin/env python3
"""Per-phase cycle breakdown of the 8192-point kernel's loop from a K1W_TIMING=1 build (probe build, timing only).

    tools/ab_build.sh "k1wtime:-DK1W_TIMING=1"
    gpurun -- 'FOSPHOR_AMD_LIB=$PWD/build/ab/lib_k1wtime.so FOSPHOR_AMD_K1_TIMING=1 FOSPHOR_AMD_OVERLAP=0 python3 tools/k1w_phase_timing.py'
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

n, batch, overlap = 8192, 16384, 2
hop = n // overlap
f = gr_fosphor_amd.Fosphor(fft_len_log=13, n_bins=512, max_spectra=batch, max_batches=1)
iq = torch.empty(((batch - 1) * hop + n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
for _ in range(4):
    assert f.process_device_overlap(iq, 1, batch, overlap) == 0
    f.finish()
L = f.L
L.fosphor_amd_debug_k1_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(256 * 2 * 16, np.int64)
assert L.fosphor_amd_debug_k1_timing(f.h, out.ctypes.data, out.size) == 0
t = out.reshape(256, 2, 16).astype(np.float64)
names = ["radix 2 (prev), loop, IQ wait", "pass 1", "IQ req + stores done + EARLY piece", "barrier 1", "LATE piece", "reads arrived",
         "pass 2", "stores done + EARLY piece", "barrier 2", "LATE piece", "reads arrived",
         "pass 3", "half stores done + EARLY piece", "barrier 3", "LATE piece", "half reads arrived"]
spw = batch / 256.0		# spectra per work-group
print("s_memtime ticks per spectrum (mean over 256 work-groups): wave 0 (early) / wave 4 (late)")
for i, nm in enumerate(names):
    print("  %-38s %8.1f %8.1f" % (nm, t[:, 0, i].mean() / spw, t[:, 1, i].mean() / spw))
print("  %-38s %8.1f %8.1f" % ("total", t[:, 0, :].sum(1).mean() / spw, t[:, 1, :].sum(1).mean() / spw))
