#!/usr/bin/env python3
"""Per-phase cycle breakdown of K1 from a K1_TIMING=1 build (debug only).

    hipcc ... -DK1_TIMING=1 -o build/ab/lib_timing.so ...      (tools/ab_build.sh "timing:-DK1_TIMING=1" builds it)
    FOSPHOR_AMD_LIB=build/ab/lib_timing.so FOSPHOR_AMD_K1_TIMING=1 FOSPHOR_AMD_OVERLAP=0 python3 tools/k1_phase_timing.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

F, B = 64, 1024
f = gr_fosphor_amd.Fosphor(n_bins=256, max_spectra=F * B, max_batches=F)
iq = torch.empty((F * B * 1024, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
OVL = int(os.environ.get("TIMING_OVERLAP", "0"))     # > 0: overlapped windows (hop = 1024 / OVL samples): cache-hot IQ
for _ in range(3):
    if OVL:
        assert f.process_device_overlap(iq, F, B, OVL) == 0
    else:
        assert f.process_device(iq, F, B) == 0
f.finish()
L = f.L
L.fosphor_amd_debug_k1_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
n_waves = 4 * int(os.environ.get("FOSPHOR_AMD_K1_BLOCKS", "512"))
out = np.zeros(n_waves * 8, np.int64)
assert L.fosphor_amd_debug_k1_timing(f.h, out.ctypes.data, out.size) == 0
t = out.reshape(n_waves, 8).astype(np.float64)
spectra_per_wave = F * B / n_waves
names = ["window(+IQ wait)+prefetch issue", "pass1+exchange", "pass2+exchange", "pass3+exchange", "pass4",
         "4th epilogue+", "epilogue", "loop/stores"]
start, end = t[:, 3].copy(), t[:, 5].copy()       # wall_clock64 (100 MHz) at wave entry / exit
t[:, 3] = 0
t[:, 5] = 0
t0 = start.min()
print("wave lifetimes on the common 100 MHz clock (us after the first wave's entry):")
print("  entry: min %.1f  median %.1f  p99 %.1f  max %.1f" % tuple(np.percentile((start - t0) / 100.0, [0, 50, 99, 100])))
print("  exit : min %.1f  median %.1f  p99 %.1f  max %.1f" % tuple(np.percentile((end - t0) / 100.0, [0, 50, 99, 100])))
ex = ((end - t0) / 100.0).reshape(-1, 4)          # [work-group][wave]
print("  exit by blockIdx %% 8 (XCD under round-robin dispatch): " + " ".join("%.1f" % ex[k::8].mean() for k in range(8)))
print("  exit of work-groups 0..255 vs 256..511 (first / second resident on a CU): %.1f / %.1f" % (ex[:256].mean(), ex[256:].mean()))
hist, edges = np.histogram(ex.ravel(), bins=12)
print("  exit histogram: " + " ".join("%.0f:%d" % (edges[i], hist[i]) for i in range(len(hist))))
tot = t.sum(1).mean()
print("cycles per spectrum per wave (mean over %d waves, %d spectra each, last launch):" % (n_waves, spectra_per_wave))
for i, nm in enumerate(names):
    print("  %-34s %8.0f  (%4.1f%%)" % (nm, t[:, i].mean() / spectra_per_wave, 100 * t[:, i].mean() / tot))
print("  %-34s %8.0f" % ("total", tot / spectra_per_wave))
