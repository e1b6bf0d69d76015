#!/usr/bin/env python3
"""Informational throughput of the non-headline BASELINE configurations on one GPU (not the bench line):

    C3  8192-point FFT, 50 % overlap fused into the read, batch 4096, 512 bins
    C5  65536-point FFT, fp16 IQ, 512 bins, the per-GPU share of a sharded frame (128 spectra)

    python3 tools/bench_configs.py
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402


def run(name, make, submit, samples_per_call, reps=20):
    f, d = make()
    for _ in range(3):
        assert submit(f, d) == 0
    f.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert submit(f, d) == 0
    f.finish()
    dt = (time.perf_counter() - t0) / reps
    print("%-4s %8.1f us per call  %9.0f MSamples/s FFT'd" % (name, dt * 1e6, samples_per_call / dt / 1e6))
    f.profile(1)
    for _ in range(5):
        assert submit(f, d) == 0
    ms, n = f.kernel_times()
    print("     K1 %.1f us  K2 %.1f us  K3 %.1f us per call" % tuple(1e3 * ms[i] / max(1, n[i]) for i in range(3)))
    f.close()


def c3():
    n, b = 8192, 4096
    f = gr_fosphor_amd.Fosphor(fft_len_log=13, n_bins=512, max_spectra=b, max_batches=8)
    d = torch.empty(((b - 1) * n // 2 + n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
    return f, d


C5_B = int(os.environ.get("C5_SPECTRA", "128"))	# per-GPU share of a frame; 128 = 1024 spectra over 8 GPUs


def c5():
    n, b = 65536, C5_B
    f = gr_fosphor_amd.Fosphor(fft_len_log=16, n_bins=512, max_spectra=b, max_batches=8, iq_fp16=True)
    d = torch.empty((b * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05).to(torch.float16)
    return f, d


if __name__ == "__main__":
    run("C3", c3, lambda f, d: f.process_device_overlap(d, 1, 4096, 2), 4096 * 8192)
    run("C5", c5, lambda f, d: f.process_device(d, 1, C5_B), C5_B * 65536)
