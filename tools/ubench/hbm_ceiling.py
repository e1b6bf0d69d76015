#!/usr/bin/env python3
"""Streaming-read / copy ceiling of the box through PyTorch's own kernels (informational: what a pure streaming kernel reaches
on this part, next to the 8 TB/s the roofline is priced against).  python3 tools/ubench/hbm_ceiling.py"""
import time
import torch

n = 1 << 30                      # 4 GiB of fp32: far beyond the 256 MiB Infinity Cache
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
torch.cuda.synchronize()


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, fn, nbytes in (("sum (read 4 GiB)", lambda: x.sum(), 4 * n),
                         ("abs-max (read 4 GiB)", lambda: x.abs().max() if False else torch.amax(x), 4 * n),
                         ("copy (read 4 + write 4 GiB)", lambda: y.copy_(x), 8 * n),
                         ("fill (write 4 GiB)", lambda: y.fill_(1.0), 4 * n)):
    dt = timed(fn)
    print("%-30s %7.1f us  %6.2f TB/s" % (name, dt * 1e6, nbytes / dt / 1e12))
