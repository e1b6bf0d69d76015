#!/usr/bin/env python3
"""One-off large-sample parity check (not part of the test suite: ~10^9 samples): hit counts of every batch
against the oracle, bit for bit, for several seeds / signal mixes / power ranges, at 128 and 256 bins.

    python3 tests/soak_parity.py [n_seeds [c2|c3|c5|pipe|share]]          (kept under tests/: it uses the oracle as the checker)
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


def big(n_seeds, log2n, fp16):
    """the 16-bit-index geometries: N = 8192 (C3) or 65536 (C5, fp16 IQ), 512 bins, device-resident input"""
    build_oracle(ref=False)
    threads = min(os.cpu_count() or 1, 64)
    n = 1 << log2n
    spectra = (1 << 22) // n * 4		# 16 Mi samples per call
    total = mismatched = 0
    t0 = time.time()
    for seed in range(n_seeds):
        rng = np.random.default_rng(5000 + seed)
        sigma = [0.05, 2e-3, 1.5, 0.3][seed % 4]
        power = [(0, 10), (-20, 5), (10, 2), (-40, 20)][(seed // 4) % 4]
        f = gr_fosphor_amd.Fosphor(fft_len_log=log2n, n_bins=512, wf_rows=64, max_spectra=spectra, iq_fp16=fp16)
        o = Oracle(fft_len_log=log2n, n_bins=512, wf_rows=64)
        f.set_power_range(*power); o.set_power_range(*power)
        for call in range(2):
            x = (rng.standard_normal((spectra * n, 2)) * sigma).astype(np.float32)
            if call:
                x = add_tone(x, sigma * 3, 0.0377)
            if fp16:
                x = x.astype(np.float16)
            assert f.process_device(torch.from_numpy(x).cuda(), 1, spectra) == 0
            assert o.process(x.astype(np.float32), strict=False, nthreads=threads) == 0
            bad = int((f.hitcount != o.hitcount.T).sum())
            mismatched += bad
            total += spectra * n
            if bad:
                print("seed %d call %d: %d cells differ" % (seed, call, bad))
        f.close()
    print("N=%d%s: %d samples, %d mismatching hit-count cells, %.0f s" % (n, " fp16" if fp16 else "", total, mismatched, time.time() - t0))
    return 1 if mismatched else 0


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


def pipe(n_calls):
    """the device-resident path at the bench's call size: 256 batches of 1024 spectra per call, sub-launched 64 at a
    time with the K1s on alternating streams, relaxed ordering.  Only the last batch's counts of a call are visible,
    but every batch's counts feed the histogram state: final counts bit-exact, state in tolerance after every call."""
    build_oracle(ref=False)
    threads = min(os.cpu_count() or 1, 64)
    F, B = 256, 1024
    f = gr_fosphor_amd.Fosphor(n_bins=256, max_spectra=F * B, max_batches=F)
    f.set_input_ordering(False)
    o = Oracle(n_bins=256)
    rng = np.random.default_rng(4242)
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    bad_cells = bad_state = 0
    t0 = time.time()
    keep = []
    for call in range(n_calls):
        d = torch.empty((F * B * 1024, 2), dtype=torch.float32, device="cuda").normal_(0.0, [0.05, 0.5, 0.003][call % 3], generator=g)
        if call & 1:
            t = torch.arange(F * B * 1024, device="cuda", dtype=torch.float32)
            d[:, 0] += 0.2 * torch.cos(0.37 * (call + 1) * t); d[:, 1] += 0.2 * torch.sin(0.37 * (call + 1) * t)
        torch.cuda.synchronize()
        keep = [d]
        assert f.process_device(d, F, B) == 0
        x = d.cpu().numpy()
        for k in range(F):
            assert o.process(x[k * B * 1024:(k + 1) * B * 1024], nthreads=threads) == 0
        bad = int((f.hitcount != o.hitcount.T).sum())
        hist_bad = int((np.abs(f.histogram - o.histogram) > 2e-6 + 1e-4 * np.abs(o.histogram)).sum())
        live_bad = int((~np.isclose(f.spectrum[..., 1], o.spectrum[..., 1], rtol=1e-4, atol=1e-6)).sum())
        wf_bad = int((~np.isclose(f.waterfall, o.waterfall, rtol=1e-4, atol=1e-6)).sum())
        bad_cells += bad; bad_state += hist_bad + live_bad + wf_bad
        print("call %d: %d count cells, %d histogram cells, %d spectrum values, %d waterfall texels differ" % (call, bad, hist_bad, live_bad, wf_bad), flush=True)
    print("pipeline: %d samples in %d calls of %d batches, %d mismatching hit-count cells, %d state values out of tolerance, %.0f s"
          % (n_calls * F * B * 1024, n_calls, F, bad_cells, bad_state, time.time() - t0))
    f.close()
    return 1 if (bad_cells or bad_state) else 0


def share(n_pairs):
    """N = 8192 in the space-sharing form (DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8): calls of 14 batches of 1024 spectra (448 tiles: the FFT launch
    runs on 224 CUs when the previous call's count / merge kernels are still pending), issued in PAIRS without looking at the
    results in between; after each pair the last batch's counts bit-exact and the state in tolerance against the oracle,
    which has seen the same 28 batches one by one."""
    build_oracle(ref=False)
    threads = min(os.cpu_count() or 1, 64)
    n, F, B = 8192, 14, 1024
    f = gr_fosphor_amd.Fosphor(fft_len_log=13, n_bins=512, max_spectra=F * B, max_batches=F)
    o = Oracle(fft_len_log=13, n_bins=512)
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    bad_cells = bad_state = 0
    t0 = time.time()
    for pair in range(n_pairs):
        ds = []
        for call in range(2):
            d = torch.empty((F * B * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, [0.05, 0.4, 0.004][(2 * pair + call) % 3], generator=g)
            if call:
                t = torch.arange(F * B * n, device="cuda", dtype=torch.float32)
                d[:, 0] += 0.1 * torch.cos(0.21 * (pair + 1) * t); d[:, 1] += 0.1 * torch.sin(0.21 * (pair + 1) * t)
            ds.append(d)
        torch.cuda.synchronize()
        for d in ds:						# back to back: the second call's FFT launch shares the chip with the first call's tail
            assert f.process_device(d, F, B) == 0
        for d in ds:
            x = d.cpu().numpy()
            for k in range(F):
                assert o.process(x[k * B * n:(k + 1) * B * n], strict=False, nthreads=threads) == 0
        bad = int((f.hitcount != o.hitcount.T).sum())
        hist_bad = int((np.abs(f.histogram - o.histogram) > 2e-6 + 1e-4 * np.abs(o.histogram)).sum())
        live_bad = int((~np.isclose(f.spectrum[..., 1], o.spectrum[..., 1], rtol=1e-4, atol=1e-6)).sum())
        wf_bad = int((~np.isclose(f.waterfall, o.waterfall, rtol=1e-4, atol=1e-6)).sum())
        bad_cells += bad; bad_state += hist_bad + live_bad + wf_bad
        print("pair %d: %d count cells, %d histogram cells, %d spectrum values, %d waterfall texels differ" % (pair, bad, hist_bad, live_bad, wf_bad), flush=True)
    print("N=8192 shared: %d samples in %d pairs of calls of %d batches, %d mismatching hit-count cells, %d state values out of tolerance, %.0f s"
          % (n_pairs * 2 * F * B * n, n_pairs, F, bad_cells, bad_state, time.time() - t0))
    f.close()
    return 1 if (bad_cells or bad_state) else 0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    which = sys.argv[2] if len(sys.argv) > 2 else "c2"
    sys.exit(main(n) if which == "c2" else pipe(n) if which == "pipe" else share(n) if which == "share" else big(n, 13, False) if which == "c3" else big(n, 16, True))
