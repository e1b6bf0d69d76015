#!/usr/bin/env python3
"""Per-phase cycle breakdown of the 65536-point kernel's loop from a K1H_TIMING=1 build (probe build, timing only; with
-DFOSPHOR_AMD_PROBES as well, FOSPHOR_AMD_DBG_K1H=15 times the loop without memory accesses and cluster waits).

    tools/ab_build.sh "k1htime:-DK1H_TIMING=1"
    gpurun -- 'FOSPHOR_AMD_LIB=$PWD/build/ab/lib_k1htime.so FOSPHOR_AMD_K1_TIMING=1 FOSPHOR_AMD_OVERLAP=0 python3 tools/k1h_phase_timing.py'
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

total, n = 1024, 65536
f = gr_fosphor_amd.Fosphor(fft_len_log=16, n_bins=512, max_spectra=total, max_batches=1, iq_fp16=True)
iq = torch.empty((total * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05).to(torch.float16)
for _ in range(4):
    assert f.process_device(iq, 1, total) == 0
    f.finish()
L = f.L
L.fosphor_amd_debug_k1_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
waves = 8
wgs = 256
out = np.zeros(wgs * 3 * 16, np.int64)
assert L.fosphor_amd_debug_k1_timing(f.h, out.ctypes.data, out.size) == 0
t = out.reshape(wgs, 3, 16).astype(np.float64)
names = ["loop overhead / tile claim", "own look at 'intermediate read by all?'", "stores + pass 1 (next)", "wait stores acked (+IQ DMA)", "barrier + arrive",
         "transpose (+pass 2 part)", "cluster barrier: poll + wg barrier", "loads issued + pass 2 (next)", "pass 3 AB incl. load wait",
         "IQ request + pass 3 CD + xb stores", "exchange barrier", "xb loads + pass 4", "-", "epilogue", "-", "-"]
live = t[:, 0, :].sum(1) > 0		# work-groups whose cluster did work
spw = total / 32.0			# spectra per cluster (mean)
print("work-groups with work: %d of %d; s_memtime cycles per spectrum (mean over them), first / middle / last wave:" % (live.sum(), wgs))
for i, nm in enumerate(names[:14]):
    print("  %-38s %8.0f %8.0f %8.0f" % (nm, *[t[live, s, i].mean() / spw for s in range(3)]))
print("  %-38s %8.0f %8.0f %8.0f" % ("total", *[t[live, s, :].sum(1).mean() / spw for s in range(3)]))
