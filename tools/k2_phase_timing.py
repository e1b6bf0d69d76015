#!/usr/bin/env python3
"""Per-phase breakdown of the count kernel (first wave of every work-group) from a probe build.

    tools/ab_build.sh "k2time:-DFOSPHOR_AMD_PROBES -DK2_TIMING"
    gpurun -- 'FOSPHOR_AMD_LIB=$PWD/build/ab/lib_k2time.so python3 tools/k2_phase_timing.py C3'      (C2 | C3 | C5)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
if cfg == "C3":
    n, batch, nb, log2n, nbat = 8192, 4096, 512, 13, 14
    f = gr_fosphor_amd.Fosphor(fft_len_log=log2n, n_bins=nb, max_spectra=batch, max_batches=nbat)
    iq = torch.empty(((nbat * batch - 1) * (n // 2) + n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
    run = lambda: f.process_device_overlap(iq, nbat, batch, 2)
elif cfg == "C5":
    n, batch, nb, log2n, nbat = 65536, 1024, 512, 16, 1
    f = gr_fosphor_amd.Fosphor(fft_len_log=log2n, n_bins=nb, max_spectra=batch, max_batches=1, iq_fp16=True)
    iq = torch.empty((batch * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05).to(torch.float16)
    run = lambda: f.process_device(iq, 1, batch)
else:
    n, batch, nb, log2n, nbat = 1024, 1024, 256, 10, 64
    f = gr_fosphor_amd.Fosphor(fft_len_log=log2n, n_bins=nb, max_spectra=batch, max_batches=nbat)
    iq = torch.empty((nbat * batch * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
    run = lambda: f.process_device(iq, nbat, batch)
L = f.L
L.fosphor_amd_debug_k2_timing.argtypes = [C.c_void_p, C.c_int]
out = np.zeros(16, np.uint64)
for _ in range(3):
    assert run() == 0
f.finish()
assert L.fosphor_amd_debug_k2_timing(out.ctypes.data, 1) == 0
reps = 8
for _ in range(reps):
    assert run() == 0
f.finish()
assert L.fosphor_amd_debug_k2_timing(out.ctypes.data, 0) == 0
wgs = float(out[15])
names = ["zeroing issued", "barrier behind it", "counting loop (this wave)", "live-sum partials", "barrier: slowest wave", "hand-off (sparse)", "row mask stored"]
print("%s: %d work-groups; s_memtime ticks per work-group (first wave):" % (cfg, wgs))
for i, nm in enumerate(names):
    print("  %-28s %9.0f" % (nm, out[i] / wgs))
print("  %-28s %9.0f" % ("total", out[:7].sum() / wgs))
