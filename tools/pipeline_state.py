#!/usr/bin/env python3
"""Measurement aid: the C2 pipeline's rate against re-allocation of the whole instance (and of the IQ ring), in ONE process."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

F, spb, N = 256, 1024, 1024
step = F * spb * N
what = sys.argv[1] if len(sys.argv) > 1 else "inst"
tune = len(sys.argv) > 2
ring = 2
keep = []
iq = f = None
for k in range(6):
    if iq is None or what in ("iq", "both"):
        iq = torch.empty((ring * step, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
        keep.append(iq)
    if f is None or what in ("inst", "both"):
        f = gr_fosphor_amd.Fosphor(n_bins=256, max_spectra=F * spb, max_batches=F, stream=torch.cuda.current_stream().cuda_stream)
        f.set_input_ordering(False)
        keep.append(f)
    tw = ""
    if tune:
        r, b, a = f.tune_placement(iq[:64 * spb * N], 64, spb)
        tw = " tune %.0f->%.0f(%d)" % (b, a, r)

    def run(n):
        for i in range(n):
            f.process_device(iq[(i % ring) * step:], F, spb)
    run(40); f.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(40); f.finish(); dt = time.perf_counter() - t0
    print("%s %d: %.0f MS/s%s" % (what, k, 40 * step / dt / 1e6, tw), flush=True)
