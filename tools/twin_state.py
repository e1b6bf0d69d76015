#!/usr/bin/env python3
"""Measurement aid: K1's memory twin (its loads and stores, no arithmetic: 98 us per 512 MiB launch in the good state, 109 in the bad one)
against re-allocation of the IQ buffer and / or the instance, and what fosphor_amd_tune_placement makes of it.
    python3 tools/twin_state.py [keep|free] [iq|inst|both] [tune]"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
what = sys.argv[2] if len(sys.argv) > 2 else "both"
tune = len(sys.argv) > 3
spb, sub = 1024, 64
n = sub * spb * 1024
keep = []
f = iq = None
res = []
for k in range(10):
    if f is None or what in ("inst", "both"):
        if f is not None and mode == "free":
            f.close()
        f = gr_fosphor_amd.Fosphor(n_bins=256, max_spectra=sub * spb, max_batches=sub)
        keep.append(f)
    if iq is None or what in ("iq", "both"):
        if mode == "free":
            iq = None
            gc.collect()
            torch.cuda.empty_cache()
        iq = torch.empty((n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05)
        if mode == "keep":
            keep.append(iq)
    torch.cuda.synchronize()
    if tune:
        r, b, a = f.tune_placement(iq, sub, spb)
        res.append("%.0f->%.0f(%d)" % (b, a, r))
    else:
        res.append("%.1f" % (f.traffic_twin(iq, sub, spb, reps=40) * 1e3))
print("%s %s: twin us = %s" % (mode, what, " ".join(res)))
