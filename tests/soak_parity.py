#!/usr/bin/env python3
"""One-off large-sample parity check (not part of the test suite: ~10^9 samples): hit counts of every batch
against the oracle, bit for bit, for several seeds / signal mixes / power ranges, at 128 and 256 bins.

    python3 tests/soak_parity.py [n_seeds]          (kept under tests/: it uses the oracle as the checker)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from _pkg import gr_fosphor_amd  # noqa: E402
from oracle_lib import Oracle, add_tone, build_oracle  # noqa: E402


def main(n_seeds):
    build_oracle(ref=False)
    threads = min(os.cpu_count() or 1, 64)
    total = mismatched = 0
    t0 = time.time()
    for seed in range(n_seeds):
        bins = (128, 256)[seed & 1]
        power = [(0, 10), (-20, 5), (10, 2), (-40, 20)][seed % 4]
        rng = np.random.default_rng(1000 + seed)
        sigma = [0.05, 1e-3, 3.0, 0.3][(seed // 2) % 4]
        f = gr_fosphor_amd.Fosphor(n_bins=bins)
        o = Oracle(n_bins=bins)
        f.set_power_range(*power); o.set_power_range(*power)
        for batch in range(16):
            x = (rng.standard_normal((1024 * 1024, 2)) * sigma).astype(np.float32)
            if batch & 1:
                x = add_tone(x, sigma * 4, 0.01 * (batch + 1))
            if batch == 7:
                x[12345] = np.inf; x[777, 1] = np.nan; x[4096:8192] = 0.0
            assert f.process(x) == 0
            assert o.process(x, nthreads=threads) == 0
            bad = int((f.hitcount != o.hitcount.T).sum())
            mismatched += bad
            total += x.shape[0]
            if bad:
                print("seed %d batch %d: %d cells differ" % (seed, batch, bad))
        f.close()
    print("%d samples, %d mismatching hit-count cells, %.0f s" % (total, mismatched, time.time() - t0))
    return 1 if mismatched else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 8))
