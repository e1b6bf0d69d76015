#!/usr/bin/env python3
"""Measurement aid: is the two-state behaviour of the 65536-point FFT kernel (232 / 247 us per frame) a property of the process or of an
instance's allocations?  Several instances in ONE process, the same input, each timed on its own."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

N, SPEC = 65536, 1024
torch.manual_seed(3)
ring = 4
iq = (torch.randn((ring * SPEC * N, 2), device="cuda") * 0.05).half()
insts, keep = [], []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    f = gr_fosphor_amd.Fosphor(n_bins=512, max_spectra=SPEC, max_batches=1, fft_len_log=16, iq_fp16=True)
    f.L.fosphor_amd_set_input_ordering(f.h, 0)
    insts.append(f)
    keep.append(torch.empty((37 + 13 * k) << 20, dtype=torch.uint8, device="cuda"))	# perturb the next instance's addresses
for rep in range(2):
    for k, f in enumerate(insts):
        for w in range(6):
            f.process_device(iq[(w % ring) * SPEC * N:], 1, SPEC)
        f.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for w in range(30):
            f.process_device(iq[(w % ring) * SPEC * N:], 1, SPEC)
        f.finish()
        dt = (time.perf_counter() - t0) / 30
        print("rep %d instance %d: %.1f us per frame, %.0f MS/s" % (rep, k, dt * 1e6, SPEC * N / dt / 1e6))
