"""GPU: parity of the HIP path (through the C ABI of libfosphor_amd.so) against the oracle
and the reference-generated golden fixtures.

Bars (BASELINE.json north_star):
  * integer hit counts            bit-exact
  * FFT output                    bit-exact (same arithmetic order as fft.cl, contraction off)
  * waterfall / live / max-hold   rtol 1e-4 (atol 1e-6: log10|X| crosses zero at |X| = 1)
  * persistence histogram floats  atol 2e-6 on values in [0, 1]
Nothing here reads /root/reference.
"""
import errno
import json
import os
import threading
import time

import numpy as np
import pytest

import golden_cases as gc
from oracle_lib import Oracle, canon_bits, gaussian_iq, add_tone, oracle_bins

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLD, "golden_meta.json")))

RTOL, ATOL = 1e-4, 1e-6
HIST_ATOL = 2e-6


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch


@pytest.fixture(scope="module")
def amd():
    from _pkg import gr_fosphor_amd
    gr_fosphor_amd.load()		# hard error when the HIP library is missing
    return gr_fosphor_amd


def close_float(a, b, rtol=RTOL, atol=ATOL):
    """allclose with inf == inf and nan == nan (position-wise)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    same_special = (np.isnan(a) & np.isnan(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b)))
    fin = np.isfinite(a) & np.isfinite(b)
    with np.errstate(invalid="ignore"):		# inf - inf at the positions same_special already covers
        ok = same_special | (fin & (np.abs(a - b) <= atol + rtol * np.abs(b)))
    return ok


def assert_close(a, b, what, rtol=RTOL, atol=ATOL):
    ok = close_float(a, b, rtol, atol)
    if not ok.all():
        idx = np.argwhere(~ok)[:5]
        a = np.asarray(a); b = np.asarray(b)
        msg = ", ".join("%s: %r vs %r" % (tuple(i), a[tuple(i)], b[tuple(i)]) for i in idx)
        raise AssertionError("%s: %d / %d outside tolerance; first: %s" % (what, (~ok).sum(), ok.size, msg))


# cells assert_hist_close let through because they sit at the fast-exit level: reported at the end of the session
# (tests/conftest.py prints it), so that a regression that starts leaning on the excuse is visible
EXCUSED = {"cells": 0, "where": []}


def assert_hist_close(h_gpu, h_ref, what):
    """Persistence histogram.  Cells the fast-exit rule (display.cl:237-238, hv <= 0.01 and no
    hits -> not rewritten) treats differently because the two float states straddle 0.01 by
    rounding are excused -- they are listed and must be rare."""
    d = np.abs(h_gpu - h_ref)
    bad = d > HIST_ATOL + RTOL * np.abs(h_ref)
    if bad.any():
        near_exit = np.abs(h_ref - 0.01) < 2e-4
        hard = bad & ~near_exit
        assert not hard.any(), "%s: %d cells differ (max %g)" % (what, hard.sum(), d[hard].max())
        assert bad.sum() <= max(2, bad.size // 20000), "%s: %d cells straddle the 0.01 fast-exit" % (what, bad.sum())
        EXCUSED["cells"] += int(bad.sum())
        EXCUSED["where"].append("%s: %d" % (what, int(bad.sum())))
        print("[assert_hist_close] %s: %d of %d cells excused (reference value within 2e-4 of the 0.01 fast-exit level); "
              "%d excused so far in this session" % (what, int(bad.sum()), bad.size, EXCUSED["cells"]))


def compare_state(f, o, what, wf_rows=None):
    """HIP instance f vs oracle o after the same calls."""
    assert f.waterfall_pos == o.waterfall_pos, what
    hc_gpu = f.hitcount			# [bin][x]
    hc_ref = o.hitcount.T		# oracle is [x][bin]
    assert np.array_equal(hc_gpu, hc_ref), "%s: hit counts differ in %d cells" % (what, (hc_gpu != hc_ref).sum())
    wf_g, wf_o = f.waterfall, o.waterfall
    if wf_rows is not None:
        wf_g, wf_o = wf_g[wf_rows], wf_o[wf_rows]
    assert_close(wf_g, wf_o, what + " waterfall")
    sp_g, sp_o = f.spectrum, o.spectrum
    assert np.array_equal(canon_bits(sp_g[..., 0]), canon_bits(sp_o[..., 0])), what + " vertex x"
    assert_close(sp_g[0, :, 1], sp_o[0, :, 1], what + " live")
    assert_close(sp_g[1, :, 1], sp_o[1, :, 1], what + " max-hold")
    assert_hist_close(f.histogram, o.histogram, what + " histogram")


# ---------------------------------------------------------------------------
# kernel level
# ---------------------------------------------------------------------------

def test_fft_bit_exact(amd, torch_cuda, oracle_built):
    torch = torch_cuda
    f = amd.Fosphor(max_spectra=256)
    o = Oracle()
    x = gaussian_iq(64 * 1024, 77, sigma=1.0).reshape(64, 1024, 2)
    x[5] *= 1e-3
    x[6] *= 1e3
    x[7] = 0.0
    x[8, 3, 0] = np.inf
    x[9, 900, 1] = np.nan
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    assert f.fft_device(d_in, d_out, 64) == 0
    got = d_out.cpu().numpy()
    want = Oracle.fft(x, o.window)
    assert np.array_equal(canon_bits(got), canon_bits(want)), \
        "%d words differ" % (canon_bits(got) != canon_bits(want)).sum()
    # custom window
    w = gc.blackman_harris()
    f.set_fft_window(w)
    assert f.fft_device(d_in, d_out, 64) == 0
    assert np.array_equal(canon_bits(d_out.cpu().numpy()), canon_bits(Oracle.fft(x, w)))
    f.close()


def _bin_inputs(o, n_bins):
    """Random magnitudes over 60 decades + every float around every bin edge + specials."""
    rng = np.random.default_rng(5)
    n = 1 << 21
    mag = np.exp(rng.uniform(np.log(1e-30), np.log(1e30), n))
    ph = rng.uniform(0, 2 * np.pi, n)
    rnd = np.stack([mag * np.cos(ph), mag * np.sin(ph)], 1).astype(np.float32)
    # mid-range, dense (the regime real spectra live in)
    mag = np.exp(rng.uniform(np.log(1e-4), np.log(1e3), n))
    ph = rng.uniform(0, 2 * np.pi, n)
    mid = np.stack([mag * np.cos(ph), mag * np.sin(ph)], 1).astype(np.float32)
    # bin edges: bisect on float bit patterns of h with (re, im) = (h, 0), then sweep +-64 ulps,
    # and also hit the same |X| off-axis
    hs, ho = o.histo_scale, o.histo_offset
    edges = []
    for b in range(1, n_bins):
        lo, hi = np.uint32(1), np.uint32(0x7F000000)
        while hi - lo > 1:
            mid_u = np.uint32((int(lo) + int(hi)) // 2)
            h = np.array([mid_u], np.uint32).view(np.float32)[0]
            if Oracle.bin(h, 0.0, hs, ho, n_bins) >= b:
                hi = mid_u
            else:
                lo = mid_u
        edges.append(int(hi))
    sweep = (np.array(edges, np.int64)[:, None] + np.arange(-64, 65)[None, :]).reshape(-1)
    hsw = sweep.astype(np.uint32).view(np.float32)
    on_axis = np.stack([hsw, np.zeros_like(hsw)], 1)
    c, s = np.float32(0.6), np.float32(0.8)
    off_axis = np.stack([hsw * c, hsw * s], 1).astype(np.float32)
    special = np.array([[0, 0], [np.inf, 0], [0, -np.inf], [np.nan, 1], [np.inf, np.nan], [1e-45, 0],
                        [1e-40, 1e-41], [3e38, 3e38], [1e20, 1e20], [-1e-20, 1e-20], [1, 0], [0, -1],
                        [2e19, 0], [1.8e19, 1e18]], np.float32)
    return np.concatenate([rnd, mid, on_axis, off_axis, special]).astype(np.float32)


@pytest.mark.parametrize("n_bins,power", [(128, (0, 10)), (256, (0, 10)), (128, (-20, 5)), (256, (10, 2)),
                                          (512, (0, 10)), (384, (-30, 3))])
def test_bin_exact(amd, torch_cuda, oracle_built, n_bins, power):
    torch = torch_cuda
    f = amd.Fosphor(n_bins=n_bins)
    o = Oracle(n_bins=n_bins)
    f.set_power_range(*power)
    o.set_power_range(*power)
    assert f.histo_scale == o.histo_scale and f.histo_offset == o.histo_offset
    v = _bin_inputs(o, n_bins)
    d = torch.from_numpy(v).cuda()
    d_bin = torch.empty(v.shape[0], dtype=torch.uint8 if n_bins <= 256 else torch.int16, device="cuda")	# 16-bit indices above 256 bins
    d_pwr = torch.empty(v.shape[0], dtype=torch.float32, device="cuda")
    want_bin, want_pwr = oracle_bins(v, o.histo_scale, o.histo_offset, n_bins)
    for force in ("0", "1"):
        os.environ["FOSPHOR_AMD_FORCE_EXACT_BIN"] = force
        assert f.bin_device(d, d_bin, d_pwr, v.shape[0]) == 0
        got = d_bin.cpu().numpy().astype(np.int32)
        bad = got != want_bin
        assert not bad.any(), "force=%s: %d bins differ, e.g. %r -> gpu %d oracle %d" % (
            force, bad.sum(), v[np.argmax(bad)], got[np.argmax(bad)], want_bin[np.argmax(bad)])
        assert_close(d_pwr.cpu().numpy(), want_pwr, "pwr (force=%s)" % force)
    os.environ.pop("FOSPHOR_AMD_FORCE_EXACT_BIN", None)
    f.close()


# ---------------------------------------------------------------------------
# whole path through fosphor_process(), against the reference fixtures
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("name", [n for n in gc.CASES if gc.CASES[n]["store"] == "full"])
def test_process_matches_reference_fixture(amd, torch_cuda, name):
    spec = gc.CASES[name]
    z = np.load(os.path.join(GOLD, name + ".npz"))
    f = amd.Fosphor()
    if "power_range" in spec:
        f.set_power_range(*spec["power_range"])
    if "window" in spec:
        f.set_fft_window(spec["window"]())
    for k in range(len(META[name]["calls"])):
        m = META[name]["calls"][k]
        pre = "c%d_" % k
        assert f.process(z[pre + "x"]) == 0
        assert f.draw() == m["pos1"]
        assert np.array_equal(f.hitcount, z[pre + "hc"].T), "%s call %d: hit counts" % (name, k)
        assert_close(f.waterfall[z[pre + "wf_idx"]], z[pre + "wf_rows"], "%s call %d waterfall" % (name, k))
        sp = f.spectrum
        assert_close(sp[0, :, 1], z[pre + "spec"][0, :, 1], "%s call %d live" % (name, k))
        assert_close(sp[1, :, 1], z[pre + "spec"][1, :, 1], "%s call %d max-hold" % (name, k))
        assert np.array_equal(canon_bits(sp[..., 0]), canon_bits(z[pre + "spec"][..., 0]))
        assert_hist_close(f.histogram, z[pre + "hist"], "%s call %d histogram" % (name, k))
    f.close()


def test_untouched_waterfall_rows_keep_noise_floor(amd, torch_cuda):
    f = amd.Fosphor()
    assert f.process(gaussian_iq(16 * 1024, 1)) == 0
    wf = f.waterfall
    assert np.all(wf[16:] == np.float32(-f.histo_offset))	# cl.c:406-433: noise floor = -power.offset
    f.close()


def test_process_argument_errors(amd, torch_cuda):
    """cl.c:882-886 error behaviour through the C ABI."""
    f = amd.Fosphor()
    assert f.process(np.zeros((17 * 1024, 2), np.float32)) == -errno.EINVAL
    assert f.process(np.zeros((1040 * 1024, 2), np.float32)) == -errno.EINVAL
    assert f.waterfall_pos == 0
    assert f.finish() == 1		# boot fills pending (cl.c:981-995)
    assert f.finish() == 0		# nothing pending (cl.c:977-979)
    assert f.process(np.zeros((16 * 1024, 2), np.float32)) == 0
    assert f.finish() == 1
    f.close()


@pytest.mark.parametrize("n_bins", [128, 256])
def test_stateful_sequence_vs_oracle(amd, torch_cuda, oracle_built, n_bins):
    """Six calls of mixed batch sizes with a drifting tone: state carry-over, ring advance."""
    f = amd.Fosphor(n_bins=n_bins)
    o = Oracle(n_bins=n_bins)
    t0 = 0
    for k, b in enumerate([16, 48, 1024, 32, 512, 64]):
        x = add_tone(gaussian_iq(b * 1024, 100 + k), 0.1, 0.05 + 0.01 * k, t0=t0)
        t0 += b * 1024
        assert f.process(x) == 0 and o.process(x, nthreads=8) == 0
        compare_state(f, o, "call %d (batch %d, %d bins)" % (k, b, n_bins))
    f.close()


def test_multi_batch_launch_equals_sequential_calls(amd, torch_cuda, oracle_built):
    """fosphor_amd_process_device(n_batches, batch) == n_batches fosphor_process calls."""
    torch = torch_cuda
    nbat, b = 5, 64
    x = add_tone(gaussian_iq(nbat * b * 1024, 42), 0.3, 0.2)
    f = amd.Fosphor(max_spectra=nbat * b)
    o = Oracle()
    d = torch.from_numpy(x).cuda()
    assert f.process_device(d, nbat, b) == 0
    for k in range(nbat):
        assert o.process(x[k * b * 1024:(k + 1) * b * 1024], nthreads=8) == 0
    compare_state(f, o, "5 x 64")
    # a second launch continues from the carried state, and wraps the ring (5*64*2 = 640 < 1024: add more)
    x2 = gaussian_iq(8 * 128 * 1024, 43)
    f2 = amd.Fosphor(max_spectra=8 * 128)
    o2 = Oracle()
    assert f2.process_device(torch.from_numpy(x2).cuda(), 8, 128) == 0
    for k in range(8):
        assert o2.process(x2[k * 128 * 1024:(k + 1) * 128 * 1024], nthreads=8) == 0
    compare_state(f2, o2, "8 x 128 (full ring)")
    f.close(); f2.close()


@pytest.mark.parametrize("env", [{"FOSPHOR_AMD_PIPE3": "1"}, {"FOSPHOR_AMD_OVERLAP": "0"}, {"FOSPHOR_AMD_K1": "2"},
                                 {"FOSPHOR_AMD_ALT": "0"}, {"FOSPHOR_AMD_TILE": "16"}, {"FOSPHOR_AMD_SUB_LOG2": "17"},
                                 {"FOSPHOR_AMD_SUB_LOG2": "17", "_relaxed": "1"},
                                 {"FOSPHOR_AMD_SUB_LOG2": "17", "FOSPHOR_AMD_PIPE3": "1"}])
def test_pipeline_options_do_not_change_results(amd, torch_cuda, oracle_built, monkeypatch, env):
    """The third stream (K3 beside the next K2, second hit-count set), the single-stream mode, the two-waves-per-spectrum
    FFT kernel (the one odd overlap hops use), the tile length, sub-launches of one batch on alternating FFT streams (with and
    without stream ordering against the caller) are scheduling choices: several back-to-back launches, then a switch to the
    sharded path and back, must leave exactly the state of the sequential reference calls."""
    torch = torch_cuda
    for k, v in env.items():
        if not k.startswith("_"):
            monkeypatch.setenv(k, v)
    nbat, b, launches = 4, 128, 4
    f = amd.Fosphor(max_spectra=nbat * b)
    if env.get("_relaxed"):
        assert f.set_input_ordering(False) == 0
    keep = []			# relaxed ordering: the caller keeps the sample buffers until finish()
    o = Oracle()
    for L in range(launches):
        x = add_tone(gaussian_iq(nbat * b * 1024, 300 + L), 0.2, 0.05 * (L + 1))
        keep.append(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        assert f.process_device(keep[-1], nbat, b) == 0		# not synchronised in between
        for k in range(nbat):
            assert o.process(x[k * b * 1024:(k + 1) * b * 1024], nthreads=8) == 0
    compare_state(f, o, "4 launches of 4 x 128 (%s)" % env)
    # sharded path right behind the pipelined launches (one rank holding the whole batch), then back
    x = gaussian_iq(256 * 1024, 310)
    assert f.accumulate_device(torch.from_numpy(x).cuda(), 256, 0, 256) == 0
    assert f.merge(256) == 0
    assert o.process(x, nthreads=8) == 0
    x = gaussian_iq(2 * 64 * 1024, 311)
    assert f.process_device(torch.from_numpy(x).cuda(), 2, 64) == 0
    for k in range(2):
        assert o.process(x[k * 64 * 1024:(k + 1) * 64 * 1024], nthreads=8) == 0
    compare_state(f, o, "after switching paths (%s)" % env)
    f.close()


def test_traffic_twin_hook_leaves_results_alone(amd, torch_cuda, oracle_built):
    """fosphor_amd_traffic_twin (K1's loads and stores without its arithmetic, for the bench's practical
    ceiling) scribbles on the intermediate buffers only: the state sequence is unaffected."""
    torch = torch_cuda
    nbat, b = 4, 64
    f = amd.Fosphor(max_spectra=nbat * b)
    o = Oracle()
    for L in range(2):
        x = gaussian_iq(nbat * b * 1024, 400 + L)
        d = torch.from_numpy(x).cuda()
        assert f.process_device(d, nbat, b) == 0
        ms = f.traffic_twin(d, nbat, b, reps=3)
        assert 0.0 < ms < 50.0
        for k in range(nbat):
            assert o.process(x[k * b * 1024:(k + 1) * b * 1024], nthreads=8) == 0
    compare_state(f, o, "launch, twin, launch, twin")
    import ctypes as C
    bad = C.c_float()
    assert f.L.fosphor_amd_traffic_twin(f.h, None, nbat, b, 3, C.byref(bad)) == -errno.EINVAL
    assert f.L.fosphor_amd_traffic_twin(f.h, d.data_ptr(), nbat + 1, b, 3, C.byref(bad)) == -errno.EINVAL	# over capacity
    f.close()


def test_ring_overwrite_within_one_launch(amd, torch_cuda, oracle_built):
    """More spectra than waterfall rows in one launch: the last wf_rows spectra survive, exactly
    as after the equivalent sequence of reference calls."""
    torch = torch_cuda
    x = gaussian_iq(3 * 512 * 1024, 44)
    f = amd.Fosphor(max_spectra=3 * 512)
    o = Oracle()
    f.process(gaussian_iq(16 * 1024, 45)); o.process(gaussian_iq(16 * 1024, 45))	# offset the ring
    assert f.process_device(torch.from_numpy(x).cuda(), 3, 512) == 0
    for k in range(3):
        o.process(x[k * 512 * 1024:(k + 1) * 512 * 1024], nthreads=8)
    compare_state(f, o, "3 x 512 over a 1024-row ring")
    f.close()


def test_big_batch_digest_fixture(amd, torch_cuda):
    """One display launch with fft_batch = 8192 (the multi-GPU semantics) against the
    reference-kernel fixture."""
    torch = torch_cuda
    z = np.load(os.path.join(GOLD, "c8_b8192.npz"))
    x = gc.CASES["c8_b8192"]["calls"]()[0]
    f = amd.Fosphor(max_spectra=8192)
    assert f.process_device(torch.from_numpy(x).cuda(), 1, 8192) == 0
    assert f.finish() == 1
    assert_hist_close(f.histogram, z["c0_hist"], "b8192 histogram")
    assert_close(f.spectrum[0, :, 1], z["c0_spec"][0, :, 1], "b8192 live")
    assert_close(f.spectrum[1, :, 1], z["c0_spec"][1, :, 1], "b8192 max-hold")
    assert_close(f.waterfall[z["c0_wf_row_sample_idx"]], z["c0_wf_row_sample"], "b8192 waterfall sample")
    hc = f.hitcount
    assert np.all(hc.sum(0) == 8192)
    f.close()


def test_ring_wrap_fixture_b512_b1024(amd, torch_cuda):
    """The only reference-generated fixture with a 1024-spectrum batch AND a ring wrap (512 spectra, then 1024
    starting at row 512: rows 512..1023 then 0..511): final hit counts bit-exact, histogram / spectrum /
    sampled waterfall rows in tolerance, ring position, through fosphor_process()."""
    name = "c5_wrap_b512_b1024"
    z = np.load(os.path.join(GOLD, name + ".npz"))
    calls = gc.CASES[name]["calls"]()
    f = amd.Fosphor()
    for k, x in enumerate(calls):
        assert f.process(x) == 0
        assert f.draw() == META[name]["calls"][k]["pos1"]
    assert np.array_equal(f.hitcount, z["c1_hc"].T), "hit counts of the wrapped 1024-spectrum batch"
    assert_hist_close(f.histogram, z["c1_hist"], "wrap histogram")
    sp = f.spectrum
    assert_close(sp[0, :, 1], z["c1_spec"][0, :, 1], "wrap live")
    assert_close(sp[1, :, 1], z["c1_spec"][1, :, 1], "wrap max-hold")
    assert np.array_equal(canon_bits(sp[..., 0]), canon_bits(z["c1_spec"][..., 0]))
    assert_close(f.waterfall[z["c1_wf_row_sample_idx"]], z["c1_wf_row_sample"], "wrap waterfall rows")
    f.close()


GLIBC_CELL_BUDGET_PER_MI = 8		# differing (bin, column) cells per 2^20 samples


@pytest.mark.parametrize("name", ["c1_gauss_b16", "c2_tone_b32x3", "c5_wrap_b512_b1024", "c6_range_m20_5"])
def test_informational_glibc_bound_reference(amd, torch_cuda, name):
    """INFORMATIONAL, with a stated budget.  The reference's kernels call OpenCL built-ins whose results are
    implementation-defined (native_sin/cos, log10, hypot: display.cl:136, fft.cl:66-67); the bit-exact fixtures
    bind them to include/fosphor_portable_math.h.  Here the same kernels were run with glibc's
    sinf/cosf/hypotf/log10f/roundf instead (tests/golden/glibc_binding_hc.npz, oracle/gen_golden.py glibc):
    the GPU's hit counts may differ from THAT run only where a sample sits within an ulp of a bin edge --
    at most GLIBC_CELL_BUDGET_PER_MI cells per 2^20 samples (measured: 4 in the 1024-spectrum call, 0 elsewhere),
    and every column still sums to the batch."""
    zg = np.load(os.path.join(GOLD, "glibc_binding_hc.npz"))
    spec = gc.CASES[name]
    f = amd.Fosphor()
    if "power_range" in spec:
        f.set_power_range(*spec["power_range"])
    for k, x in enumerate(spec["calls"]()):
        assert f.process(x) == 0
        hc = f.hitcount.astype(np.int64)
        ref = zg["%s_c%d_hc" % (name, k)].T.astype(np.int64)
        n = x.shape[0]
        differ = int((hc != ref).sum())
        budget = max(GLIBC_CELL_BUDGET_PER_MI, GLIBC_CELL_BUDGET_PER_MI * n // (1 << 20))
        assert differ <= budget, "%s call %d: %d cells differ from the glibc-bound run (budget %d)" % (name, k, differ, budget)
        assert np.abs(hc - ref).max() <= 2 and np.all(hc.sum(0) == n // 1024)
    f.close()


def test_n1024_with_512_bins_whole_path(amd, torch_cuda, oracle_built):
    """N = 1024 with more than 256 bins needs 16-bit bin indices: the general kernel at 128 threads per
    spectrum (every process entry point used to fail with -EIO on such an instance)."""
    torch = torch_cuda
    f = amd.Fosphor(n_bins=512, max_spectra=256)
    o = Oracle(n_bins=512)
    assert f.histo_scale == o.histo_scale
    t0 = 0
    for k, b in enumerate([32, 64, 256]):
        x = add_tone(gaussian_iq(b * 1024, 600 + k), 0.15, 0.07 + 0.03 * k, t0=t0)
        t0 += b * 1024
        if k == 1:
            assert f.process_device(torch.from_numpy(x).cuda(), 2, b // 2) == 0
            for h in range(2):
                assert o.process(x[h * (b // 2) * 1024:(h + 1) * (b // 2) * 1024], nthreads=8) == 0
        else:
            assert f.process(x) == 0 and o.process(x, nthreads=8) == 0
        compare_state(f, o, "N=1024 / 512 bins, call %d" % k)
    f.close()


def test_c3_full_batch_properties(amd, torch_cuda, oracle_built):
    """BASELINE config C3 at its full size -- 8192-point FFT, 50 % overlap fused into the read, ONE batch of
    4096 spectra, 512 bins -- through size-independent properties: every column's counts sum to the batch;
    the counts of the batch are the sum of the counts of its two 2048-spectrum halves (separate instances,
    each reading its half of the same unexpanded stream); two runs give identical bits; and the last 64
    spectra's waterfall rows equal the oracle's for exactly those spectra."""
    torch = torch_cuda
    n, nb, over, B = 8192, 512, 2, 4096
    hop = n // over
    x = add_tone(gaussian_iq((B - 1) * hop + n, 777), 0.05, 0.0313)
    d = torch.from_numpy(x).cuda()
    f = amd.Fosphor(fft_len_log=13, n_bins=nb, max_spectra=B)
    assert f.process_device_overlap(d, 1, B, over) == 0
    hc = f.hitcount.astype(np.int64)
    assert np.all(hc.sum(0) == B)
    halves = []
    for h in range(2):
        g = amd.Fosphor(fft_len_log=13, n_bins=nb, max_spectra=B)
        assert g.process_device_overlap(d[h * (B // 2) * hop:], 1, B // 2, over) == 0
        halves.append(g.hitcount.astype(np.int64))
        g.close()
    assert np.array_equal(hc, halves[0] + halves[1]), "hit counts are not additive over time blocks"
    f2 = amd.Fosphor(fft_len_log=13, n_bins=nb, max_spectra=B)
    assert f2.process_device_overlap(d, 1, B, over) == 0
    assert np.array_equal(f2.hitcount, f.hitcount)
    assert np.array_equal(canon_bits(f2.histogram), canon_bits(f.histogram))
    assert np.array_equal(canon_bits(f2.waterfall), canon_bits(f.waterfall))
    assert np.array_equal(canon_bits(f2.spectrum), canon_bits(f.spectrum))
    # oracle slice: the last 64 spectra of the batch as a batch of their own (waterfall rows are stateless)
    o = Oracle(fft_len_log=13, n_bins=nb)
    first = B - 64
    expanded = np.concatenate([x[(first + i) * hop:(first + i) * hop + n] for i in range(64)])
    assert o.process(expanded, strict=False, nthreads=8) == 0
    rows_gpu = (f.waterfall_pos - 64 + np.arange(64)) & 1023
    rows_ref = (o.waterfall_pos - 64 + np.arange(64)) & 1023
    assert_close(f.waterfall[rows_gpu], o.waterfall[rows_ref], "C3 batch 4096: last 64 waterfall rows")
    f.close(); f2.close()


@pytest.mark.parametrize("over,F,sub_log2", [(2, 14, None), (4, 28, "27")])
def test_c3_space_sharing_is_bit_identical(amd, torch_cuda, monkeypatch, over, F, sub_log2):
    """N = 8192, space sharing (DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8): a call whose tiles are a multiple of 224 -- here 14 batches of 1024
    spectra = 448 tiles of 32 -- runs its FFT kernel on 224 work-groups and the count / merge kernels of the PREVIOUS launch on
    the CUs it leaves free.  Two such calls back to back (the second call's FFT kernel beside the first call's count and merge)
    must leave exactly the state of the single-stream form (fosphor_amd_set_overlap(0): 256 work-groups, one kernel at a
    time): same kernels, same order per cell, so every buffer bit for bit; and every column's counts sum to the batch.
    Second case: 28 batches per call cut into two pieces of 14 (sub-launches of 128 Mi samples), 75 % overlap."""
    torch = torch_cuda
    if sub_log2:
        monkeypatch.setenv("FOSPHOR_AMD_SUB_LOG2", sub_log2)
    n, nb, B = 8192, 512, 1024
    hop = n // over
    xs = [add_tone(gaussian_iq((F * B - 1) * hop + n, 4242 + k), 0.05, 0.0313 + 0.01 * k) for k in range(2)]
    ds = [torch.from_numpy(x).cuda() for x in xs]
    res = []
    for shared in (True, False):
        f = amd.Fosphor(fft_len_log=13, n_bins=nb, max_spectra=F * B, max_batches=F)
        if not shared:
            assert f.set_overlap(False) == 0
        for d in ds:
            assert f.process_device_overlap(d, F, B, over) == 0
        assert f.finish() == 1
        res.append((f.hitcount.copy(), f.histogram.copy(), f.waterfall.copy(), f.spectrum.copy(), f.waterfall_pos))
        f.close()
    a, b = res
    assert np.all(a[0].astype(np.int64).sum(0) == B)
    assert np.array_equal(a[0], b[0]), "hit counts differ between the shared and the single-stream form"
    assert np.array_equal(canon_bits(a[1]), canon_bits(b[1])), "histogram"
    assert np.array_equal(canon_bits(a[2]), canon_bits(b[2])), "waterfall"
    assert np.array_equal(canon_bits(a[3]), canon_bits(b[3])), "spectrum"
    assert a[4] == b[4]


def test_buffers_without_hitcount_view(amd, torch_cuda, oracle_built, monkeypatch):
    """fosphor_amd_get_buffers_nohc: the pointers / ring position / scale a front end polls per frame, without the hit-count view
    (no export kernel, no wait).  The view made afterwards is still that of the last batch -- also with the single-stream
    pipeline (FOSPHOR_AMD_OVERLAP=0), where the count kernel does not run on the count/merge stream."""
    torch = torch_cuda
    for overlap in ("1", "0"):
        monkeypatch.setenv("FOSPHOR_AMD_OVERLAP", overlap)
        f = amd.Fosphor(max_spectra=256)
        o = Oracle()
        x = add_tone(gaussian_iq(4 * 64 * 1024, 61), 0.1, 0.2)
        assert f.process_device(torch.from_numpy(x).cuda(), 4, 64) == 0
        b0 = f.buffers(hitcount=False)			# nothing synchronised yet
        assert not b0.d_hitcount and b0.d_histogram and b0.d_waterfall and b0.d_spectrum
        assert b0.waterfall_pos == 256 and (b0.fft_len, b0.n_bins, b0.wf_rows) == (1024, 128, 1024)
        b1 = f.buffers()				# makes the view, behind the count kernel that wrote the 16-bit counts
        assert b1.d_hitcount and b1.d_histogram == b0.d_histogram and b1.waterfall_pos == b0.waterfall_pos
        from gr_fosphor_amd.dist import wrap_device_array
        hc = wrap_device_array(b1.d_hitcount, (128, 1024), torch.int32).cpu().numpy().view(np.uint32)	# the view is complete when buffers() returns
        for k in range(4):
            assert o.process(x[k * 64 * 1024:(k + 1) * 64 * 1024]) == 0
        assert np.array_equal(hc, o.hitcount.T), "overlap=%s" % overlap
        f.close()


def test_table_change_between_relaxed_calls(amd, torch_cuda, oracle_built, monkeypatch):
    """Relaxed input ordering: the FFT kernels of a call may still be running on the second FFT stream when the next call
    begins.  A window / power-range change between two such calls must not reach the earlier call's spectra (the tables are
    re-uploaded only after every FFT stream has drained) and must apply to all of the later call's: the state equals the
    oracle's after the same sequence."""
    torch = torch_cuda
    monkeypatch.setenv("FOSPHOR_AMD_SUB_LOG2", "17")		# 2 batches of 64 spectra per sub-launch: 4 sub-launches per call
    nbat, b = 8, 64
    f = amd.Fosphor(max_spectra=nbat * b)
    assert f.set_input_ordering(False) == 0
    o = Oracle()
    rng = np.random.default_rng(5)
    keep = []
    for call in range(4):
        x = add_tone(gaussian_iq(nbat * b * 1024, 700 + call), 0.1, 0.07 * (call + 1))
        keep.append(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()
        if call in (1, 3):
            win = (0.5 + rng.random(1024)).astype(np.float32)		# a window that visibly changes every bin
            f.set_fft_window(win); o.set_window(win)
        if call == 2:
            f.set_power_range(-10, 5); o.set_power_range(-10, 5)
        assert f.process_device(keep[-1], nbat, b) == 0			# no synchronisation between the calls
        for k in range(nbat):
            assert o.process(x[k * b * 1024:(k + 1) * b * 1024], nthreads=8) == 0
    assert f.finish() >= 0
    compare_state(f, o, "tables changed between relaxed calls")
    f.close()


def test_full_size_properties(amd, torch_cuda):
    """BASELINE config C2 size (batch 1024, 256 bins), 4 batches per launch: size-independent
    properties -- counts sum to the batch per column, determinism, and hit-count additivity
    (counts of a 2048-batch = sum of the counts of its two 1024-halves)."""
    torch = torch_cuda
    x = gaussian_iq(2048 * 1024, 46)
    d = torch.from_numpy(x).cuda()
    f = amd.Fosphor(n_bins=256, max_spectra=4096)
    assert f.process_device(d, 2, 1024) == 0
    hc_b = f.hitcount.astype(np.int64)			# second half
    assert np.all(hc_b.sum(0) == 1024)
    g = amd.Fosphor(n_bins=256, max_spectra=4096)
    assert g.process_device(d, 1, 1024) == 0
    hc_a = g.hitcount.astype(np.int64)			# first half
    h = amd.Fosphor(n_bins=256, max_spectra=4096)
    assert h.process_device(d, 1, 2048) == 0
    assert np.array_equal(h.hitcount.astype(np.int64), hc_a + hc_b)
    # determinism: same input, same launch -> identical bits everywhere
    f2 = amd.Fosphor(n_bins=256, max_spectra=4096)
    assert f2.process_device(d, 2, 1024) == 0
    assert np.array_equal(canon_bits(f.histogram), canon_bits(f2.histogram))
    assert np.array_equal(canon_bits(f.waterfall), canon_bits(f2.waterfall))
    assert np.array_equal(canon_bits(f.spectrum), canon_bits(f2.spectrum))
    for q in (f, g, h, f2):
        q.close()


def _bench_iq(torch, n_samples, seed, tone):
    """the bench's synthetic input (white complex Gaussian, sigma 0.05 per component, generated on the device), optionally
    with a tone so that the persistence state has structure; returns (device tensor, host copy)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    d = torch.empty((n_samples, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05, generator=g)
    if tone:
        t = torch.arange(n_samples, device="cuda", dtype=torch.float32)
        d[:, 0] += 0.1 * torch.cos(0.61 * t)
        d[:, 1] += 0.1 * torch.sin(0.61 * t)
    torch.cuda.synchronize()
    return d, d.cpu().numpy()


def test_bench_launch_shape_vs_oracle(amd, torch_cuda, oracle_built):
    """Hit counts bit-exact for the LAST of 512 batches only (the only counts a call exposes); the other 511 are seen through the
    float state they went into, in tolerance.  The launch shape bench.py times (BASELINE C2, the headline): 256 bins, ONE
    fosphor_amd_process_device call of 256 x 1024 spectra = 2^28 samples -- four 64-batch sub-launches on alternating FFT streams, relaxed input ordering --
    then a second such call on the other half of the input ring, queued behind the first without a host wait in between.
    Semantics: cl.c:870-968 applied 256 times per call.  The oracle takes the same 512 batches one fosphor_process at a
    time: the last batch's hit counts equal it bit for bit and the state every one of the 512 batches went into --
    persistence histogram, live, max-hold, waterfall ring, ring position -- is in tolerance."""
    torch = torch_cuda
    F, B = 256, 1024
    threads = min(os.cpu_count() or 1, 64)
    f = amd.Fosphor(n_bins=256, max_spectra=F * B, max_batches=F, stream=torch.cuda.current_stream().cuda_stream)
    f.set_input_ordering(False)				# bench.py's default: the ring is written once, before the first call
    o = Oracle(n_bins=256)
    halves = [_bench_iq(torch, F * B * 1024, 7 + h, tone=bool(h)) for h in range(2)]
    for d, _ in halves:					# both calls in flight together, as in the timed loop
        assert f.process_device(d, F, B) == 0
    assert f.finish() >= 0
    for _, x in halves:
        for k in range(F):
            assert o.process(x[k * B * 1024:(k + 1) * B * 1024], nthreads=threads) == 0
    compare_state(f, o, "bench launch shape, batch mode (2 calls x 256 batches)")
    f.close()


def test_bench_frame_mode_shape_vs_oracle(amd, torch_cuda, oracle_built):
    """`bench.py --mode frame` (what every rank of the multi-GPU bench runs; here one rank, no exchange): the 256 x 1024
    spectra of a step are ONE display frame -- accumulate_device (sub-launched K1s, count kernels taking several chunks,
    k2c_sum) then one merge with fft_batch = 262144, i.e. one reference display launch over the whole frame (the kernel is
    batch-generic, only the host caps it: cl.c:885).  Two frames back to back, against the oracle's two such launches."""
    torch = torch_cuda
    from gr_fosphor_amd.dist import ShardedFosphor
    F, B = 256, 1024
    threads = min(os.cpu_count() or 1, 64)
    sf = ShardedFosphor(amd.Fosphor, 0, 1, n_bins=256, max_spectra=F * B, max_batches=F)
    sf.f.set_input_ordering(False)
    o = Oracle(n_bins=256)
    frames = [_bench_iq(torch, F * B * 1024, 17 + h, tone=bool(h)) for h in range(2)]
    for d, _ in frames:
        sf.frame(d, F * B, overlap=True, wait_producer=False)
    sf.flush()
    assert sf.f.finish() >= 0
    for _, x in frames:
        assert o.process(x, strict=False, nthreads=threads) == 0
    f = sf.f
    assert f.waterfall_pos == o.waterfall_pos
    assert np.array_equal(f.hitcount, o.hitcount.T), "frame mode: hit counts differ"
    assert_close(f.waterfall, o.waterfall, "frame mode waterfall")
    assert_close(f.spectrum[0, :, 1], o.spectrum[0, :, 1], "frame mode live")
    assert_close(f.spectrum[1, :, 1], o.spectrum[1, :, 1], "frame mode max-hold")
    assert_hist_close(f.histogram, o.histogram, "frame mode histogram")
    sf.close()


def test_sharded_batch_equals_single_launch(amd, torch_cuda, oracle_built):
    """The multi-GPU split run on one device: two 'ranks' each accumulate half of one 2048-spectrum
    batch, the partial arrays are combined the way the all-reduce would (sum, sum, max), one merge:
    identical counts and matching floats vs the single-launch path and vs the oracle."""
    torch = torch_cuda
    total = 2048
    x = add_tone(gaussian_iq(total * 1024, 47), 0.05, 0.31)
    d = torch.from_numpy(x).cuda()
    ranks = [amd.Fosphor(max_spectra=total) for _ in range(2)]
    parts = []
    for r, fr in enumerate(ranks):
        off = r * (total // 2)
        assert fr.accumulate_device(d[off * 1024:(off + total // 2) * 1024], total // 2, off, total) == 0
        fr.finish()
        parts.append(fr.partials())
    # combine on the device with torch (stand-in for RCCL all-reduce)
    from gr_fosphor_amd.dist import wrap_device_array
    hc = [wrap_device_array(p.d_hc, (p.n_hc,), torch.int32) for p in parts]
    ls = [wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32) for p in parts]
    mx = [wrap_device_array(p.d_max, (p.n_cols,), torch.float32) for p in parts]
    hc_sum = hc[0] + hc[1]
    ls_sum = ls[0] + ls[1]
    mx_max = torch.maximum(mx[0], mx[1])
    for r in range(2):
        hc[r].copy_(hc_sum); ls[r].copy_(ls_sum); mx[r].copy_(mx_max)
    torch.cuda.synchronize()
    for fr in ranks:
        assert fr.merge(total) == 0
    o = Oracle()
    assert o.process(x, strict=False, nthreads=8) == 0
    # rank 0 holds rows of the first half only; compare hist/spectrum/counts
    for fr in ranks:
        assert np.array_equal(fr.hitcount, o.hitcount.T)
        assert_hist_close(fr.histogram, o.histogram, "sharded histogram")
        assert_close(fr.spectrum[0, :, 1], o.spectrum[0, :, 1], "sharded live")
        assert_close(fr.spectrum[1, :, 1], o.spectrum[1, :, 1], "sharded max-hold")
        assert fr.waterfall_pos == o.waterfall_pos
    # waterfall rows: rank 1 owns the last 1024 spectra = the whole surviving ring
    assert_close(ranks[1].waterfall, o.waterfall, "sharded waterfall (rank 1 rows)")
    for fr in ranks:
        fr.close()


@pytest.mark.parametrize("world,no_sum16,sliced", [(2, "", False), (2, "1", False), (4, "", False), (8, "", False),
                                                   (4, "", True), (8, "", True)])
def test_sharded_frame_many_chunks(amd, torch_cuda, oracle_built, monkeypatch, world, no_sum16, sliced):
    """BASELINE C4 in emulation: `world` instances on this one GPU take total / world spectra each of one
    8192-spectrum batch through dist.shard_range (world 8: 1024 per rank, the configuration's own split).  A shard of
    several 1024-spectrum chunks leaves per-chunk packed 16-bit slabs + k2c_sum (default) or 32-bit global atomics
    (FOSPHOR_AMD_NO_SUM16=1).  torch adds stand in for the collective: all-reduce + full merge, or (sliced) what
    reduce-scatter leaves -- rank r holds the summed counts only in cells [r C / world, (r + 1) C / world) -- followed by
    fosphor_amd_merge_sliced and an emulated all-gather of the histogram slices.  Every rank must end with the oracle's state
    for the whole batch."""
    torch = torch_cuda
    from gr_fosphor_amd.dist import shard_range, wrap_device_array
    if no_sum16:
        monkeypatch.setenv("FOSPHOR_AMD_NO_SUM16", "1")
    else:
        monkeypatch.delenv("FOSPHOR_AMD_NO_SUM16", raising=False)
    total, nb = 8192, 256
    x = add_tone(gaussian_iq(total * 1024, 53), 0.04, -0.12)
    d = torch.from_numpy(x).cuda()
    ranks = [amd.Fosphor(max_spectra=total // world, n_bins=nb) for _ in range(world)]
    parts = []
    for r, fr in enumerate(ranks):
        off, n = shard_range(total, r, world)
        assert (off, n) == (r * (total // world), total // world)
        assert fr.accumulate_device(d[off * 1024:(off + n) * 1024], n, off, total) == 0
        fr.finish()
        parts.append(fr.partials())
    hc = [wrap_device_array(p.d_hc, (p.n_hc,), torch.int32) for p in parts]
    ls = [wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32) for p in parts]
    mx = [wrap_device_array(p.d_max, (p.n_cols,), torch.float32) for p in parts]
    hc_sum = torch.stack(hc).sum(0, dtype=torch.int32)
    ls_sum = torch.stack(ls).sum(0)
    mx_max = torch.stack(mx).max(0).values
    cells = nb * 1024
    per = cells // world
    for r in range(world):
        if sliced:
            # a reduce-scatter leaves the sums in the rank's own slice only; poison the rest so that a merge reading
            # outside its slice cannot pass
            hc[r].fill_(0x5a5a5a5)
            hc[r][r * per:(r + 1) * per].copy_(hc_sum[r * per:(r + 1) * per])
        else:
            hc[r].copy_(hc_sum)
        ls[r].copy_(ls_sum); mx[r].copy_(mx_max)
    torch.cuda.synchronize()
    o = Oracle(n_bins=nb)
    assert o.process(x, strict=False, nthreads=8) == 0
    if sliced:
        for r, fr in enumerate(ranks):
            assert fr.merge_sliced(total, world, r) == 0
            assert fr.merge_sliced(total, 3, 0) == -errno.EINVAL		# cells % world != 0
            assert fr.finish() >= 0
        hists = [fr.histogram.reshape(-1) for fr in ranks]
        full = np.concatenate([hists[r][r * per:(r + 1) * per] for r in range(world)]).reshape(nb, 1024)	# the all-gather
        assert_hist_close(full, o.histogram, "frame histogram, world %d sliced" % world)
        for r, fr in enumerate(ranks):
            # outside its slice a rank's histogram is untouched (still the boot value 0)
            own = np.zeros(cells, dtype=bool); own[r * per:(r + 1) * per] = True
            assert not hists[r][~own].any(), "rank %d wrote outside its slice" % r
            assert_close(fr.spectrum[0, :, 1], o.spectrum[0, :, 1], "frame live (sliced)")
            assert_close(fr.spectrum[1, :, 1], o.spectrum[1, :, 1], "frame max-hold (sliced)")
    else:
        for fr in ranks:
            assert fr.merge(total) == 0
        for fr in ranks:
            assert np.array_equal(fr.hitcount, o.hitcount.T)
            assert int(fr.hitcount.sum()) == total * 1024
            assert_hist_close(fr.histogram, o.histogram, "frame histogram, world %d" % world)
            assert_close(fr.spectrum[0, :, 1], o.spectrum[0, :, 1], "frame live")
            assert_close(fr.spectrum[1, :, 1], o.spectrum[1, :, 1], "frame max-hold")
    # the ring holds the last 1024 spectra of the batch: with `world` ranks each owns the rows it computed
    rows_per = total // world
    for r, fr in enumerate(ranks):
        lo = max(r * rows_per, total - 1024)
        hi = (r + 1) * rows_per
        if hi > lo:
            sel = slice(lo - (total - 1024), hi - (total - 1024))
            assert_close(fr.waterfall[sel], o.waterfall[sel], "frame waterfall (rank %d rows)" % r)
    for fr in ranks:
        fr.close()


@pytest.mark.parametrize("group", ["1", "2", "4"])
def test_sharded_frame_grouped_chunks(amd, torch_cuda, oracle_built, monkeypatch, group):
    """The sub-launched frame path (a shard longer than one sub-launch): the count kernel takes `group` consecutive
    1024-spectrum chunks per work-group (FOSPHOR_AMD_FRAME_GROUP; default 4 where it divides) -- fewer 16-bit slabs
    through memory, same counts.  One rank holding the whole 8192-spectrum frame, sub-launches of 4 chunks."""
    torch = torch_cuda
    monkeypatch.setenv("FOSPHOR_AMD_SUB_LOG2", "22")
    monkeypatch.setenv("FOSPHOR_AMD_FRAME_GROUP", group)
    total = 8192
    x = add_tone(gaussian_iq(total * 1024, 57), 0.03, 0.21)
    f = amd.Fosphor(max_spectra=total, n_bins=256)
    o = Oracle(n_bins=256)
    for call in range(2):
        assert f.accumulate_device(torch.from_numpy(x).cuda(), total, 0, total) == 0
        assert f.merge(total) == 0
        assert o.process(x, strict=False, nthreads=8) == 0
        assert np.array_equal(f.hitcount, o.hitcount.T), "call %d" % call
        assert_hist_close(f.histogram, o.histogram, "grouped frame histogram")
        assert_close(f.spectrum[0, :, 1], o.spectrum[0, :, 1], "grouped frame live")
        assert_close(f.spectrum[1, :, 1], o.spectrum[1, :, 1], "grouped frame max-hold")
    assert_close(f.waterfall, o.waterfall, "grouped frame waterfall")
    f.close()


def overlap_cc_reference(x, wlen, overlap):
    """numpy restatement of lib/overlap_cc_impl.cc:64-79: windows of wlen samples whose starts
    advance wlen/overlap input samples, concatenated."""
    hop = wlen // overlap
    n_win = (x.shape[0] - wlen) // hop + 1
    return np.concatenate([x[i * hop:i * hop + wlen] for i in range(n_win)])


@pytest.mark.parametrize("overlap", [2, 4])
def test_fused_overlap_equals_materialised_stream(amd, torch_cuda, oracle_built, overlap):
    """N1: reading overlapped windows straight from the unexpanded stream == feeding the sink the
    stream overlap_cc(1024, overlap) would have produced (BASELINE config C3's 50 % overlap)."""
    torch = torch_cuda
    n_spec = 2 * 64
    hop = 1024 // overlap
    x = add_tone(gaussian_iq((n_spec - 1) * hop + 1024, 48), 0.1, 0.07)
    expanded = overlap_cc_reference(x, 1024, overlap)
    assert expanded.shape[0] == n_spec * 1024
    f = amd.Fosphor(max_spectra=n_spec)
    assert f.process_device_overlap(torch.from_numpy(x).cuda(), 2, 64, overlap) == 0
    o = Oracle()
    for k in range(2):
        assert o.process(expanded[k * 64 * 1024:(k + 1) * 64 * 1024]) == 0
    compare_state(f, o, "fused overlap %d" % overlap)
    # and against the product's own non-fused path on the materialised stream
    g = amd.Fosphor(max_spectra=n_spec)
    assert g.process_device(torch.from_numpy(expanded).cuda(), 2, 64) == 0
    assert np.array_equal(f.hitcount, g.hitcount)
    assert np.array_equal(canon_bits(f.waterfall), canon_bits(g.waterfall))
    assert np.array_equal(canon_bits(f.histogram), canon_bits(g.histogram))
    assert f.process_device_overlap(torch.from_numpy(x).cuda(), 2, 64, 3) == -errno.EINVAL
    f.close(); g.close()


@pytest.mark.parametrize("seed,relaxed", [(1, False), (2, True), (3, False), (4, True)])
def test_random_call_sequences(amd, torch_cuda, oracle_built, monkeypatch, seed, relaxed):
    """Random mixes of every entry point -- fosphor_process, device-resident calls of 1..12 batches (cut into sub-launches
    of 64 spectra here, K1s alternating between the two FFT streams, waterfall rings flipping whenever a call rewrites
    the 64-row ring), fused-overlap calls, sharded frames (accumulate + merge), draws in between -- against the oracle fed
    the same spectra in the same order."""
    torch = torch_cuda
    monkeypatch.setenv("FOSPHOR_AMD_SUB_LOG2", "16")
    rng = np.random.default_rng(9000 + seed)
    n = 1024
    f = amd.Fosphor(n_bins=256, wf_rows=64, max_spectra=2048, max_batches=16)
    if relaxed:
        assert f.set_input_ordering(False) == 0
    o = Oracle(n_bins=256, wf_rows=64)
    keep = []
    t0 = 0
    for step in range(14):
        kind = rng.integers(0, 5)
        if kind == 0:					# reference entry point, host samples
            b = int(rng.choice([16, 32, 64, 160]))
            x = add_tone(gaussian_iq(b * n, 9100 + 20 * seed + step), 0.1, 0.03 * (step + 1), t0=t0)
            assert f.process(x) == 0 and o.process(x, nthreads=8) == 0
        elif kind in (1, 2):				# device-resident call, several batches
            nb, b = int(rng.integers(1, 13)), int(rng.choice([16, 48, 128]))
            x = add_tone(gaussian_iq(nb * b * n, 9100 + 20 * seed + step), 0.1, 0.03 * (step + 1), t0=t0)
            keep.append(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
            assert f.process_device(keep[-1], nb, b) == 0
            for k in range(nb):
                assert o.process(x[k * b * n:(k + 1) * b * n], nthreads=8) == 0
        elif kind == 3:					# overlap_cc fused into the read
            nb, b, over = int(rng.integers(1, 5)), 64, int(rng.choice([2, 4]))
            hop = n // over
            x = add_tone(gaussian_iq((nb * b - 1) * hop + n, 9100 + 20 * seed + step), 0.1, 0.021 * (step + 1), t0=t0)
            keep.append(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
            assert f.process_device_overlap(keep[-1], nb, b, over) == 0
            ex = overlap_cc_reference(x, n, over)
            for k in range(nb):
                assert o.process(ex[k * b * n:(k + 1) * b * n], nthreads=8) == 0
        else:						# a sharded frame held by one rank
            b = int(rng.choice([64, 256, 2048]))
            x = add_tone(gaussian_iq(b * n, 9100 + 20 * seed + step), 0.1, 0.017 * (step + 1), t0=t0)
            keep.append(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
            assert f.accumulate_device(keep[-1], b, 0, b) == 0 and f.merge(b) == 0
            assert o.process(x, strict=False, nthreads=8) == 0
        t0 += 4096
        if rng.integers(0, 3) == 0:
            assert f.draw() == o.waterfall_pos
            compare_state(f, o, "seed %d step %d (kind %d)" % (seed, step, kind))
    compare_state(f, o, "seed %d, end" % seed)
    f.close()


def test_sharded_frame_with_fused_overlap(amd, torch_cuda, oracle_built):
    """fosphor_amd_accumulate_device_overlap: two time shards of one 256-spectrum frame, each reading its part of the
    UNEXPANDED stream (overlap 2), combined the way the exchange would: the counts, and after the merge the state, of
    the single launch over the materialised stream (oracle)."""
    torch = torch_cuda
    from gr_fosphor_amd.dist import combine_partials_numpy, wrap_device_array
    n, over, B = 1024, 2, 256
    hop = n // over
    x = add_tone(gaussian_iq((B - 1) * hop + n, 471), 0.1, 0.171)
    expanded = overlap_cc_reference(x, n, over)[:B * n]
    o = Oracle()
    assert o.process(expanded, strict=False, nthreads=8) == 0
    d = torch.from_numpy(x).cuda()
    parts = []
    ranks = []
    for r in range(2):
        f = amd.Fosphor(max_spectra=B)
        off = r * (B // 2)
        assert f.accumulate_device(d[off * hop:], B // 2, off, B, overlap=over) == 0
        assert f.finish() >= 0
        p = f.partials()
        parts.append((wrap_device_array(p.d_hc, (p.n_hc,), torch.int32).cpu().numpy().view(np.uint32),
                      wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32).cpu().numpy(),
                      wrap_device_array(p.d_max, (p.n_cols,), torch.float32).cpu().numpy()))
        ranks.append(f)
    hc, live, vmax = combine_partials_numpy(parts)
    assert np.array_equal(hc.reshape(128, 1024), o.hitcount.T), "sharded overlap counts"
    f = ranks[0]
    p = f.partials()
    wrap_device_array(p.d_hc, (p.n_hc,), torch.int32).copy_(torch.from_numpy(hc.view(np.int32)))
    wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32).copy_(torch.from_numpy(live))
    wrap_device_array(p.d_max, (p.n_cols,), torch.float32).copy_(torch.from_numpy(vmax))
    torch.cuda.synchronize()
    assert f.merge(B) == 0
    assert_hist_close(f.histogram, o.histogram, "sharded overlap histogram")
    assert_close(f.spectrum[0, :, 1], o.spectrum[0, :, 1], "sharded overlap live")
    assert_close(f.spectrum[1, :, 1], o.spectrum[1, :, 1], "sharded overlap max-hold")
    assert f.accumulate_device(d, B, 0, B, overlap=3) == -errno.EINVAL
    for q in ranks:
        q.close()


def test_sink_runtime_streams_through_fifo(amd, torch_cuda, oracle_built):
    """N2: the GNU-Radio-free sink (work() -> pinned fifo -> worker thread -> fosphor_process ->
    fosphor_draw), fed like the GR scheduler feeds base_sink_c_impl::work.  Batch boundaries depend
    on thread timing (as in the reference), so the check uses what does not: every sample is
    processed once, the ring position, and the waterfall rows (one per spectrum)."""
    import ctypes as C
    L = amd.load()
    n_spec = 3000						# > 1024: wraps the ring; not a multiple of 16
    x = add_tone(gaussian_iq(n_spec * 1024, 49), 0.1, 0.21)
    s = L.fosphor_amd_sink_new()
    L.fosphor_amd_sink_set_frequency_range(s, 100e6, 2e6)
    assert L.fosphor_amd_sink_start(s) == 1
    flat = np.ascontiguousarray(x).reshape(-1)
    pos, total = 0, n_spec * 1024
    chunk = 37 * 1024 + 5					# deliberately unaligned chunks
    while pos < total:
        n = min(chunk, total - pos)
        took = L.fosphor_amd_sink_work(s, flat[2 * pos:].ctypes.data, n)
        assert 0 <= took <= n
        pos += took
    L.fosphor_amd_sink_stop(s)					# drains whole 16-spectrum groups
    frames, samples, db_ref, db_div, frozen = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_int(), C.c_int()
    L.fosphor_amd_sink_stats(s, C.byref(frames), C.byref(samples), C.byref(db_ref), C.byref(db_div), C.byref(frozen))
    done_spec = (n_spec // 16) * 16
    assert samples.value == done_spec * 1024 and frames.value >= 1
    assert (db_ref.value, db_div.value, frozen.value) == (0, 10, 0)
    L.fosphor_amd_sink_free(s)

    # the same stream through a plain instance in one go, as the reference for the rows
    s2 = L.fosphor_amd_sink_new()
    L.fosphor_amd_sink_ui_action(s2, 0)			# DB_PER_DIV_UP: 10 -> 20 dB/div
    L.fosphor_amd_sink_ui_action(s2, 3)			# REF_DOWN by 20
    L.fosphor_amd_sink_ui_action(s2, 11)			# FREEZE_TOGGLE
    L.fosphor_amd_sink_stats(s2, None, None, C.byref(db_ref), C.byref(db_div), C.byref(frozen))
    assert (db_ref.value, db_div.value, frozen.value) == (-20, 20, 1)
    L.fosphor_amd_sink_free(s2)


def test_sink_waterfall_matches_oracle(amd, torch_cuda, oracle_built):
    """Rows written through the streaming sink equal the oracle's rows for the same stream."""
    import ctypes as C
    L = amd.load()
    n_spec = 1600
    x = gaussian_iq(n_spec * 1024, 50)
    s = L.fosphor_amd_sink_new()
    assert L.fosphor_amd_sink_start(s) == 1
    flat = np.ascontiguousarray(x).reshape(-1)
    pos = 0
    while pos < n_spec * 1024:
        pos += L.fosphor_amd_sink_work(s, flat[2 * pos:].ctypes.data, min(64 * 1024, n_spec * 1024 - pos))
    # read the result through the core BEFORE stop() releases it: wait until everything is consumed
    import time
    samples = C.c_uint64()
    for _ in range(2000):
        L.fosphor_amd_sink_stats(s, None, C.byref(samples), None, None, None)
        if samples.value == n_spec * 1024:
            break
        time.sleep(0.005)
    assert samples.value == n_spec * 1024
    core = L.fosphor_amd_sink_core(s)
    wf = np.empty((1024, 1024), np.float32)
    assert L.fosphor_amd_read(core, 0, wf.ctypes.data, wf.nbytes) == 0
    b = amd.Buffers(); L.fosphor_amd_get_buffers(core, C.byref(b))
    assert b.waterfall_pos == n_spec & 1023
    o = Oracle()
    for k in range(0, n_spec, 400):
        o.process(x[k * 1024:(k + 400) * 1024], nthreads=8)
    assert_close(wf, o.waterfall, "sink waterfall")
    L.fosphor_amd_sink_stop(s)
    L.fosphor_amd_sink_free(s)


def test_sink_zoom_pane_and_click_to_frequency(amd, torch_cuda):
    """base_sink_c_impl.cc:257-296,371-397: window reshape, the 65 % / 35 % split when the zoom pane is on, and a click
    turned into the frequency under the cursor -- against the same layout rules applied through the public
    fosphor_render_* / fosphor_pos2freq API (whose arithmetic is pinned against the reference's fosphor.c by
    tests/test_render_geometry.py)."""
    import ctypes as C
    import time
    L = amd.load()
    s = L.fosphor_amd_sink_new()
    L.fosphor_amd_sink_set_frequency_range(s, 433.92e6, 2.0e6)
    L.fosphor_amd_sink_reshape(s, 1280, 720)
    assert L.fosphor_amd_sink_start(s) == 1
    x = gaussian_iq(64 * 1024, 52)
    L.fosphor_amd_sink_work(s, np.ascontiguousarray(x).ctypes.data, 64 * 1024)		# one frame: settings applied
    samples = C.c_uint64()
    for _ in range(400):
        L.fosphor_amd_sink_stats(s, None, C.byref(samples), None, None, None)
        if samples.value == 64 * 1024:
            break
        time.sleep(0.005)
    core = L.fosphor_amd_sink_core(s)
    r = amd.Render()
    L.fosphor_amd_sink_get_render(s, 0, C.byref(r))
    assert (r.width, r.height) == (1280, 720) and (r.options & (1 << 7)) == 0		# FRO_CHANNELS off without zoom
    freq = C.c_double()
    mid_y = int(r._y_histo[0] + 5)
    assert L.fosphor_amd_sink_mouse_action(s, 0, 640, mid_y, C.byref(freq)) == 1
    assert freq.value == L.fosphor_pos2freq(core, C.byref(r), 640)
    assert abs(freq.value - 433.92e6) < 2.0e6
    assert L.fosphor_amd_sink_mouse_action(s, 0, 5000, mid_y, C.byref(freq)) == 0	# outside every pane

    L.fosphor_amd_sink_ui_action(s, 4)							# ZOOM_TOGGLE
    L.fosphor_amd_sink_work(s, np.ascontiguousarray(x).ctypes.data, 64 * 1024)
    for _ in range(400):
        L.fosphor_amd_sink_stats(s, None, C.byref(samples), None, None, None)
        if samples.value == 2 * 64 * 1024:
            break
        time.sleep(0.005)
    time.sleep(0.05)
    z = amd.Render()
    L.fosphor_amd_sink_get_render(s, 0, C.byref(r))
    L.fosphor_amd_sink_get_render(s, 1, C.byref(z))
    a = int(1280 * np.float32(0.65))
    assert r.width == a and z.pos_x == a - 10 and z.width == 1280 - a + 10 and z.height == 720
    assert r.channels[0].enabled == 1 and abs(z.freq_span - 0.2) < 1e-7 and abs(z.freq_center - 0.5) < 1e-7
    xz = int(z._x[0] + 0.5 * (z._x[1] - z._x[0]))
    yz = int(z._y_histo[0] + 5)
    assert L.fosphor_amd_sink_mouse_action(s, 0, xz, yz, C.byref(freq)) == 1
    assert freq.value == L.fosphor_pos2freq(core, C.byref(z), xz)
    assert abs(freq.value - 433.92e6) < 0.2 * 2.0e6					# the zoom pane spans 20 % around the centre

    # the "freq" callback runs with no lock of the sink held (the reference publishes its message unlocked,
    # base_sink_c_impl.cc:385,390): a callback that re-enters the sink must not deadlock
    seen = []
    CB = C.CFUNCTYPE(None, C.c_double, C.c_void_p)

    def on_freq(fv, _user):
        rr = amd.Render()
        L.fosphor_amd_sink_get_render(s, 1, C.byref(rr))			# takes the render lock
        inner = C.c_double()
        depth = len(seen)
        seen.append(fv)
        if depth == 0:								# one level of re-entry through the click path itself
            assert L.fosphor_amd_sink_mouse_action(s, 0, xz, yz, C.byref(inner)) == 1
            assert inner.value == fv

    cb = CB(on_freq)
    L.fosphor_amd_sink_set_freq_callback(s, C.cast(cb, C.c_void_p), None)
    done = []
    th = threading.Thread(target=lambda: done.append(L.fosphor_amd_sink_mouse_action(s, 0, xz, yz, C.byref(freq))))
    th.start()
    th.join(20.0)
    assert not th.is_alive(), "execute_mouse_action deadlocked on a re-entrant callback"
    assert done == [1] and len(seen) == 2 and seen[0] == seen[1] == freq.value
    L.fosphor_amd_sink_set_freq_callback(s, None, None)
    L.fosphor_amd_sink_stop(s)
    L.fosphor_amd_sink_free(s)


# PCIe-inclusive floor for the streaming sink fed through work() (GSamples/s; 8 B per sample over a Gen5 x16 link:
# 7.9 GSamples/s is the bound).  FOSPHOR_SINK_FLOOR overrides it on a loaded or slower host.
SINK_WORK_FLOOR = float(os.environ.get("FOSPHOR_SINK_FLOOR", "4.5"))
# (round 4, idle box, 16 Mi-sample FIFO: 6.3 GSamples/s with the zero-copy feed, 5.8-6.4 through work() over 256 Mi samples
# (tools/sink_bench.py), 5.0-5.3 over the 36 Mi samples of the test below, start-up included)


def test_sink_zero_copy_feed(amd, torch_cuda, oracle_built):
    """The zero-copy producer interface: the source writes its samples straight into the pinned FIFO
    (write_prepare / write_commit) and the host copy of work() disappears.  Same results as work(); and the rate of
    the sink alone -- regions committed as they are -- is reported (PCIe-bound)."""
    import ctypes as C
    L = amd.load()
    n_spec = 2048 + 512
    x = add_tone(gaussian_iq(n_spec * 1024, 53), 0.1, 0.21)
    flat = np.ascontiguousarray(x).reshape(-1)
    s = L.fosphor_amd_sink_new_len(1 << 24)
    got = C.c_int()
    assert L.fosphor_amd_sink_write_prepare(s, 1 << 16, C.byref(got), 10) is None and got.value == 0	# not running
    assert L.fosphor_amd_sink_start(s) == 1
    pos, total = 0, n_spec * 1024
    while pos < total:
        p = L.fosphor_amd_sink_write_prepare(s, min(1 << 19, total - pos), C.byref(got), 1000)
        assert p and got.value > 0
        dst = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(got.value * 2,))
        dst[:] = flat[2 * pos:2 * (pos + got.value)]				# "the driver's receive call"
        L.fosphor_amd_sink_write_commit(s, got.value)
        pos += got.value
    samples = C.c_uint64()
    for _ in range(2000):
        L.fosphor_amd_sink_stats(s, None, C.byref(samples), None, None, None)
        if samples.value == total:
            break
        time.sleep(0.005)
    assert samples.value == total
    core = L.fosphor_amd_sink_core(s)
    wf = np.empty((1024, 1024), np.float32)
    assert L.fosphor_amd_read(core, 0, wf.ctypes.data, wf.nbytes) == 0
    o = Oracle()
    for k in range(0, n_spec, 512):
        o.process(x[k * 1024:(k + 512) * 1024], nthreads=8)
    assert_close(wf, o.waterfall, "sink waterfall after a zero-copy feed")
    # rate of the sink alone: commit regions as they lie (whatever the ring holds), 1 Mi samples at a time
    t0 = time.perf_counter()
    n_rate = 0
    while n_rate < (1 << 30):
        p = L.fosphor_amd_sink_write_prepare(s, 1 << 20, C.byref(got), 1000)
        assert p
        L.fosphor_amd_sink_write_commit(s, got.value)
        n_rate += got.value
    want = total + (n_rate & ~(16 * 1024 - 1))
    while samples.value < want:
        L.fosphor_amd_sink_stats(s, None, C.byref(samples), None, None, None)
        assert time.perf_counter() - t0 < 60.0
    rate = n_rate / (time.perf_counter() - t0) / 1e9
    print("sink, zero-copy feed: %.2f GSamples/s (no host copy; PCIe Gen5 x16 bound 7.9)" % rate)
    assert rate > SINK_WORK_FLOOR
    L.fosphor_amd_sink_stop(s)
    L.fosphor_amd_sink_free(s)


def test_placement_tuning_leaves_results_alone(amd, torch_cuda, oracle_built):
    """fosphor_amd_tune_placement re-allocates intermediate sets (the FFT kernel's memory traffic runs in one of two states, decided by
    the allocations involved); whatever it replaces, the state that has been accumulated and the results of later calls are the
    oracle's.  It reports sane times and refuses what the traffic twin cannot time."""
    import errno
    torch = torch_cuda
    L = amd.load()
    f = amd.Fosphor(n_bins=256, max_spectra=16 * 1024, max_batches=16)
    o = Oracle(n_bins=256)
    x = add_tone(gaussian_iq(4 * 1024 * 1024, 123), 0.15, 0.11).reshape(4, 1024 * 1024, 2)
    d = torch.from_numpy(x).cuda()
    big = torch.randn((16 * 1024 * 1024, 2), device="cuda") * 0.05			# what the tuning times its traffic against (128 MiB and more: below, it only measures)
    assert f.process_device(d[:2], 2, 1024) == 0
    for k in range(2):
        assert o.process(x[k], nthreads=8) == 0
    replaced, before, after = f.tune_placement(big, 16, 1024, max_tries=3)		# between two calls: the accumulated state must survive
    print("placement tuning: %d sets replaced, slowest set %.1f -> %.1f us per launch" % (replaced, before, after))
    assert 0 <= replaced <= 3 * 2 and 0.0 < after <= before * 1.02 and before < 1000.0
    small = f.tune_placement(d[:2], 2, 1024, max_tries=3)				# a short launch: measured, nothing replaced
    assert small[0] == 0
    assert f.process_device(d[2:], 2, 1024) == 0
    for k in range(2, 4):
        assert o.process(x[k], nthreads=8) == 0
    f.draw()
    compare_state(f, o, "across a placement tuning")
    assert L.fosphor_amd_tune_placement(f.h, big.data_ptr(), 32, 1024, 4, None, None) == -errno.EINVAL	# more than max_spectra
    assert L.fosphor_amd_tune_placement(f.h, big.data_ptr(), 4, 1024, 0, None, None) == -errno.EINVAL
    f.close()
    g = amd.Fosphor(n_bins=512, max_spectra=1024)						# 16-bit indices: the twin does not model them
    assert L.fosphor_amd_tune_placement(g.h, big.data_ptr(), 1, 1024, 2, None, None) == -errno.EINVAL
    g.close()


def test_upload_and_kernels_in_two_steps(amd, torch_cuda, oracle_built):
    """fosphor_amd_upload_pinned / fosphor_amd_process_uploaded (what the sink's frame loop calls): two uploads may be pending, a
    third is refused with -EBUSY, the kernels are queued oldest first, and a call that carries several whole batches is applied like
    so many calls.  State against the oracle after every step."""
    import ctypes as C
    import errno
    torch = torch_cuda
    L = amd.load()
    f = amd.Fosphor(max_spectra=3072)
    o = Oracle()
    x = add_tone(gaussian_iq(5 * 1024 * 1024 + 48 * 1024, 97), 0.2, 0.37)
    pin = torch.from_numpy(x).pin_memory()
    base = pin.data_ptr()
    a, b = 48 * 1024, 1024 * 1024
    assert L.fosphor_amd_pending_uploads(f.h) == 0
    assert L.fosphor_amd_process_uploaded(f.h, None) == -errno.EINVAL
    assert L.fosphor_amd_upload_pinned(f.h, base, a) == 0				# 48 spectra
    assert L.fosphor_amd_upload_pinned(f.h, base + 8 * a, b) == 0			# one batch
    assert L.fosphor_amd_pending_uploads(f.h) == 2
    assert L.fosphor_amd_upload_pinned(f.h, base + 8 * (a + b), b) == -errno.EBUSY
    assert L.fosphor_amd_upload_pinned(f.h, base, 1024 * 1024 + 16 * 1024) == -errno.EINVAL	# beyond a batch: whole batches only
    got = C.c_int()
    assert L.fosphor_amd_process_uploaded(f.h, C.byref(got)) == 0 and got.value == a
    assert o.process(x[:a]) == 0
    assert L.fosphor_amd_process_uploaded(f.h, C.byref(got)) == 0 and got.value == b
    assert o.process(x[a:a + b], nthreads=8) == 0
    f.draw()
    compare_state(f, o, "two pending uploads")
    # three whole batches in one call == three calls (the middle one through process_pinned, which takes pending uploads first)
    assert L.fosphor_amd_upload_pinned(f.h, base + 8 * (a + b), 3 * b) == 0
    assert L.fosphor_amd_process_pinned(f.h, base + 8 * (a + 4 * b), b) == 0
    assert L.fosphor_amd_pending_uploads(f.h) == 0
    for k in range(1, 5):
        assert o.process(x[a + k * b:a + (k + 1) * b], nthreads=8) == 0
    f.draw()
    compare_state(f, o, "three batches in one call, then one")
    f.close()


def test_sink_native_feed_keeps_uploads_in_flight(amd, torch_cuda, oracle_built):
    """The sink fed from a native thread in large work() calls (helper-thread copies, several FIFO regions in
    flight, regions discarded on their own upload events): every sample is processed exactly once and the final
    rows equal the oracle's."""
    import ctypes as C
    L = amd.load()
    n_spec = 4096 + 512
    x = add_tone(gaussian_iq(n_spec * 1024, 51), 0.1, 0.33)
    s = L.fosphor_amd_sink_new_len(1 << 24)
    assert s and L.fosphor_amd_sink_new_len(1000) is None
    assert L.fosphor_amd_sink_start(s) == 1
    flat = np.ascontiguousarray(x).reshape(-1)
    dt = L.fosphor_amd_sink_feed(s, flat.ctypes.data, n_spec * 1024, 1 << 20, 1)
    assert dt > 0
    samples = C.c_uint64()
    L.fosphor_amd_sink_stats(s, None, C.byref(samples), None, None, None)
    assert samples.value == n_spec * 1024
    assert L.fosphor_amd_sink_dropped(s) == 0			# nothing was refused by the device
    core = L.fosphor_amd_sink_core(s)
    wf = np.empty((1024, 1024), np.float32)
    assert L.fosphor_amd_read(core, 0, wf.ctypes.data, wf.nbytes) == 0
    o = Oracle()
    for k in range(0, n_spec, 512):
        o.process(x[k * 1024:(k + 512) * 1024], nthreads=8)
    assert_close(wf, o.waterfall, "sink waterfall after a native feed")
    # a rate, not only a count: the same samples again, several times (PCIe-inclusive; round 3: 4.65 GSamples/s measured with a 16 Mi
    # FIFO on an idle box -- the floor here is a sixth of that, for a shared box and this 4 Mi FIFO)
    reps = 8
    dt = L.fosphor_amd_sink_feed(s, flat.ctypes.data, n_spec * 1024, 1 << 20, reps)
    assert dt > 0
    rate = reps * n_spec * 1024 / dt / 1e9
    print("sink, native feed: %.2f GSamples/s through work()" % rate)
    assert rate > SINK_WORK_FLOOR, "streaming sink moved only %.2f GSamples/s" % rate
    L.fosphor_amd_sink_stop(s)
    # a sink that is not running takes nothing and says so instead of blocking for ever (work() returns 0, the feed gives up)
    assert L.fosphor_amd_sink_work(s, flat.ctypes.data, 1 << 16) == 0
    L.fosphor_amd_sink_free(s)


# ---------------------------------------------------------------------------
# BASELINE config C3 geometry: N = 8192, 512 bins, 50 % overlap  (no reference behaviour exists
# beyond N = 1024 / 128 bins: the oracle's generalisation defines it -- parity unpinned)
# ---------------------------------------------------------------------------

def test_fft8192_bit_exact(amd, torch_cuda, oracle_built):
    torch = torch_cuda
    f = amd.Fosphor(fft_len_log=13, n_bins=512, max_spectra=64)
    o = Oracle(fft_len_log=13, n_bins=512)
    x = gaussian_iq(8 * 8192, 80, sigma=1.0).reshape(8, 8192, 2)
    x[3] *= 1e-4
    x[5, 17, 0] = np.nan
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    assert f.fft_device(d_in, d_out, 8) == 0
    want = Oracle.fft(x, o.window, fft_len_log=13)
    got = d_out.cpu().numpy()
    assert np.array_equal(canon_bits(got), canon_bits(want)), "%d words differ" % (canon_bits(got) != canon_bits(want)).sum()
    # and it is a DFT
    ref = np.fft.fft((x[0, :, 0].astype(np.float64) + 1j * x[0, :, 1]) * o.window.astype(np.float64))
    g = got[0, :, 0].astype(np.float64) + 1j * got[0, :, 1]
    assert np.max(np.abs(g - ref)) / np.max(np.abs(ref)) < 1e-6
    f.close()


@pytest.mark.parametrize("overlap,tile", [(2, None), (2, "32"), (2, "64"), (4, "32"), (8, "64"), (16, "32"), (4, None),
                                          (1, "32"), (32, "32"), (8192, None)])
def test_c3_geometry_vs_oracle(amd, torch_cuda, oracle_built, monkeypatch, overlap, tile):
    """8192-point FFT, 512 bins, overlap_cc fused into the read, two launches with state carry-over.  The overlapped part of a
    window is reused from registers through one branch per ratio (overlap 2 / 4 / 8 / 16; anything else -- no overlap, 32, the odd
    hop of overlap 8192 -- reloads every row), at the tile lengths the real C3 launch uses (32, 64) as well as the short ones
    small launches pick."""
    torch = torch_cuda
    n, nb = 8192, 512
    hop = n // overlap
    if tile:
        monkeypatch.setenv("FOSPHOR_AMD_TILE", tile)
    f = amd.Fosphor(fft_len_log=13, n_bins=nb, max_spectra=128)
    o = Oracle(fft_len_log=13, n_bins=nb)
    assert f.histo_scale == o.histo_scale and f.histo_offset == o.histo_offset
    t0 = 0
    for call, n_spec in enumerate([64, 128] if tile == "64" else [32, 64]):
        x = add_tone(gaussian_iq((n_spec - 1) * hop + n, 81 + call), 0.05, 0.0313, t0=t0)
        t0 += x.shape[0]
        expanded = np.concatenate([x[i * hop:i * hop + n] for i in range(n_spec)])
        assert f.process_device_overlap(torch.from_numpy(x).cuda(), 1, n_spec, overlap) == 0
        assert o.process(expanded, strict=False, nthreads=8) == 0
        assert f.waterfall_pos == o.waterfall_pos
        assert np.array_equal(f.hitcount, o.hitcount.T), "call %d: hit counts" % call
        rows = (o.waterfall_pos - n_spec + np.arange(n_spec)) & 1023
        assert_close(f.waterfall[rows], o.waterfall[rows], "C3 waterfall")
        assert_close(f.spectrum[0, :, 1], o.spectrum[0, :, 1], "C3 live")
        assert_close(f.spectrum[1, :, 1], o.spectrum[1, :, 1], "C3 max-hold")
        assert_hist_close(f.histogram, o.histogram, "C3 histogram")
    # host path: len must be a multiple of 16 * 8192 here
    assert f.process(np.zeros((16 * 1024, 2), np.float32)) == -errno.EINVAL
    assert f.process(gaussian_iq(16 * 8192, 83)) == 0 and o.process(gaussian_iq(16 * 8192, 83)) == 0
    assert np.array_equal(f.hitcount, o.hitcount.T)
    f.close()


def test_fft65536_bit_exact(amd, torch_cuda, oracle_built):
    """N = 65536 in two 256-point levels (passes p = 1, 16 per residue mod 256 inside a wavefront; passes p = 256, 4096 per
    offset mod 256 in a work-group; the spectrum between them in the XCD's L2): the bits of the oracle's radix-16 plan of FMA butterflies
    (this build's own plan at this length: no reference behaviour exists), fp32 and fp16 input."""
    torch = torch_cuda
    n = 65536
    o = Oracle(fft_len_log=16, n_bins=512, wf_rows=64)
    x = gaussian_iq(8 * n, 90, sigma=1.0).reshape(8, n, 2)
    x[3] *= 1e-4
    x[5, 40000, 1] = np.inf
    x[6, 123, 0] = np.nan
    f = amd.Fosphor(fft_len_log=16, n_bins=512, wf_rows=64, max_spectra=16)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    assert f.fft_device(d_in, d_out, 8) == 0
    want = Oracle.fft(x, o.window, fft_len_log=16)
    got = d_out.cpu().numpy()
    assert np.array_equal(canon_bits(got), canon_bits(want)), "%d words differ" % (canon_bits(got) != canon_bits(want)).sum()
    ref = np.fft.fft((x[0, :, 0].astype(np.float64) + 1j * x[0, :, 1]) * o.window.astype(np.float64))
    g = got[0, :, 0].astype(np.float64) + 1j * got[0, :, 1]
    assert np.max(np.abs(g - ref)) / np.max(np.abs(ref)) < 2e-6
    f.close()
    # fp16 IQ: widened exactly on load
    xh = (x[:4] * 0.05).astype(np.float16)
    fh = amd.Fosphor(fft_len_log=16, n_bins=512, wf_rows=64, max_spectra=16, iq_fp16=True)
    d_h = torch.from_numpy(xh).cuda()
    d_o = torch.empty((4, n, 2), dtype=torch.float32, device="cuda")
    assert fh.fft_device(d_h, d_o, 4) == 0
    want = Oracle.fft(xh.astype(np.float32), o.window, fft_len_log=16)
    assert np.array_equal(canon_bits(d_o.cpu().numpy()), canon_bits(want))
    fh.close()


def test_c5_geometry_vs_oracle(amd, torch_cuda, oracle_built):
    """BASELINE config C5 on one GPU: 65536-point FFT, fp16 IQ, 512 bins; two launches with state
    carry-over, then the host path.  The oracle gets the same fp16 values widened to fp32."""
    torch = torch_cuda
    n, nb, rows = 65536, 512, 64
    f = amd.Fosphor(fft_len_log=16, n_bins=nb, wf_rows=rows, max_spectra=64, iq_fp16=True)
    o = Oracle(fft_len_log=16, n_bins=nb, wf_rows=rows)
    assert f.histo_scale == o.histo_scale and f.histo_offset == o.histo_offset
    t0 = 0
    for call, n_spec in enumerate([16, 48]):
        x = add_tone(gaussian_iq(n_spec * n, 91 + call), 0.05, 0.0313, t0=t0).astype(np.float16)
        t0 += n_spec * n
        assert f.process_device(torch.from_numpy(x).cuda(), 1, n_spec) == 0
        assert o.process(x.astype(np.float32), strict=False, nthreads=8) == 0
        assert f.waterfall_pos == o.waterfall_pos
        assert np.array_equal(f.hitcount, o.hitcount.T), "call %d: hit counts" % call
        assert int(f.hitcount.sum()) == n_spec * n
        keep = (o.waterfall_pos - min(n_spec, rows) + np.arange(min(n_spec, rows))) & (rows - 1)
        assert_close(f.waterfall[keep], o.waterfall[keep], "C5 waterfall")
        assert_close(f.spectrum[0, :, 1], o.spectrum[0, :, 1], "C5 live")
        assert_close(f.spectrum[1, :, 1], o.spectrum[1, :, 1], "C5 max-hold")
        assert_hist_close(f.histogram, o.histogram, "C5 histogram")
    xh = gaussian_iq(16 * n, 93).astype(np.float16)
    assert f.process(xh) == 0 and o.process(xh.astype(np.float32), strict=False) == 0	# host path, fp16 samples
    assert np.array_equal(f.hitcount, o.hitcount.T)
    assert f.process(xh[:8 * n]) == -errno.EINVAL						# not a multiple of 16 spectra
    f.close()
    # fp16 IQ with the 1024-point kernels is refused at init, loudly
    with pytest.raises(RuntimeError):
        amd.Fosphor(iq_fp16=True)


@pytest.mark.parametrize("world,sliced", [(2, False), (8, True)])
def test_c5_sharded_two_ranks(amd, torch_cuda, oracle_built, world, sliced):
    """C5 sharded (BASELINE configs[4] in emulation): `world` instances on this GPU take total / world spectra each of one
    frame (fp16 IQ).  world 2: all-reduce + full merge on both.  world 8: what ShardedFosphor does for this 128 MiB state --
    reduce-scatter (rank r keeps the summed counts of cells [r C / 8, (r + 1) C / 8) only), fosphor_amd_merge_sliced, and the
    all-gather of the histogram slices -- with torch copies standing in for RCCL.  Combined state = the oracle's for the frame."""
    torch = torch_cuda
    from gr_fosphor_amd.dist import shard_range, wrap_device_array
    n, nb, rows = 65536, 512, 64
    total = 64 if world == 2 else 128
    x = add_tone(gaussian_iq(total * n, 95), 0.03, 0.2).astype(np.float16)
    d = torch.from_numpy(x).cuda()
    ranks = [amd.Fosphor(fft_len_log=16, n_bins=nb, wf_rows=rows, max_spectra=total // world, max_batches=2, iq_fp16=True)
             for _ in range(world)]
    parts = []
    for r, fr in enumerate(ranks):
        off, cnt = shard_range(total, r, world)
        assert fr.accumulate_device(d[off * n:(off + cnt) * n], cnt, off, total) == 0
        fr.finish()
        parts.append(fr.partials())
    hc = [wrap_device_array(p.d_hc, (p.n_hc,), torch.int32) for p in parts]
    ls = [wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32) for p in parts]
    mx = [wrap_device_array(p.d_max, (p.n_cols,), torch.float32) for p in parts]
    hc_sum = torch.stack(hc).sum(0, dtype=torch.int32)
    ls_sum = torch.stack(ls).sum(0)
    mx_max = torch.stack(mx).max(0).values
    cells = nb * n
    per = cells // world
    for r in range(world):
        if sliced:
            hc[r].fill_(0x5a5a5a5)
            hc[r][r * per:(r + 1) * per].copy_(hc_sum[r * per:(r + 1) * per])
        else:
            hc[r].copy_(hc_sum)
        ls[r].copy_(ls_sum); mx[r].copy_(mx_max)
    torch.cuda.synchronize()
    o = Oracle(fft_len_log=16, n_bins=nb, wf_rows=rows)
    assert o.process(x.astype(np.float32), strict=False, nthreads=8) == 0
    if sliced:
        for r, fr in enumerate(ranks):
            assert fr.merge_sliced(total, world, r) == 0
        hists = [fr.histogram.reshape(-1) for fr in ranks]
        full = np.concatenate([hists[r][r * per:(r + 1) * per] for r in range(world)]).reshape(nb, n)
        assert_hist_close(full, o.histogram, "C5 sharded histogram (8 slices)")
        assert np.array_equal(hc_sum.cpu().numpy().view(np.uint32).reshape(nb, n), o.hitcount.T)
        for fr in ranks:
            assert_close(fr.spectrum[0, :, 1], o.spectrum[0, :, 1], "C5 sharded live")
            assert_close(fr.spectrum[1, :, 1], o.spectrum[1, :, 1], "C5 sharded max-hold")
    else:
        for fr in ranks:
            assert fr.merge(total) == 0
        for fr in ranks:
            assert np.array_equal(fr.hitcount, o.hitcount.T)
            assert_hist_close(fr.histogram, o.histogram, "C5 sharded histogram")
            assert_close(fr.spectrum[0, :, 1], o.spectrum[0, :, 1], "C5 sharded live")
            assert_close(fr.spectrum[1, :, 1], o.spectrum[1, :, 1], "C5 sharded max-hold")
    # a rank owns the ring rows of the spectra it computed among the last `rows` of the frame
    per_rank = total // world
    for r, fr in enumerate(ranks):
        lo, hi = max(r * per_rank, total - rows), (r + 1) * per_rank
        if hi > lo:
            sel = [(t & (rows - 1)) for t in range(lo, hi)]
            assert_close(fr.waterfall[sel], o.waterfall[sel], "C5 sharded waterfall (rank %d rows)" % r)
    for fr in ranks:
        fr.close()


def test_c5_full_frame_properties(amd, torch_cuda, oracle_built):
    """BASELINE C5 at the size bench.py --config C5 times: ONE frame of 1024 spectra x 65536 points (64 Mi samples of fp16
    IQ, 512 bins) through the default (fused) FFT kernel and the sparse count hand-off.  The oracle cannot run 64 Mi samples
    in seconds, so: size-independent properties (every column's counts sum to 1024; same input -> same bits), additivity
    against two smaller launches, one of which -- the frame's first 16 spectra -- is checked against the oracle bit for bit,
    and those 16 spectra's waterfall rows of the full frame against the oracle's."""
    torch = torch_cuda
    n, nb, rows, total = 65536, 512, 1024, 1024
    g = torch.Generator(device="cuda"); g.manual_seed(97)
    d = torch.empty((total * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, 0.05, generator=g).to(torch.float16)
    kw = dict(fft_len_log=16, n_bins=nb, wf_rows=rows, max_batches=2, iq_fp16=True)
    f = amd.Fosphor(max_spectra=total, **kw)
    assert f.process_device(d, 1, total) == 0
    assert f.finish() >= 0
    hc = f.hitcount.astype(np.int64)
    assert np.all(hc.sum(0) == total), "a column's counts do not sum to the frame length"
    wf = f.waterfall
    hist, spec = canon_bits(f.histogram), canon_bits(f.spectrum)
    # determinism: a second instance, same frame -> identical bits everywhere
    f2 = amd.Fosphor(max_spectra=total, **kw)
    assert f2.process_device(d, 1, total) == 0 and f2.finish() >= 0
    assert np.array_equal(f2.hitcount, hc)
    assert np.array_equal(canon_bits(f2.waterfall), canon_bits(wf))
    assert np.array_equal(canon_bits(f2.histogram), hist) and np.array_equal(canon_bits(f2.spectrum), spec)
    f2.close()
    # additivity: counts of the frame = counts of its first 16 spectra + counts of the other 1008
    a = amd.Fosphor(max_spectra=16, **kw)
    b = amd.Fosphor(max_spectra=1008, **kw)
    assert a.process_device(d[:16 * n], 1, 16) == 0 and b.process_device(d[16 * n:], 1, 1008) == 0
    hc_a, hc_b = a.hitcount.astype(np.int64), b.hitcount.astype(np.int64)
    assert np.array_equal(hc, hc_a + hc_b), "hit counts are not additive over the frame's spectra"
    # the 16-spectrum slice against the oracle: counts bit-exact, its rows of the FULL frame's waterfall in tolerance
    x16 = d[:16 * n].cpu().numpy().astype(np.float32)
    o = Oracle(fft_len_log=16, n_bins=nb, wf_rows=rows)
    assert o.process(x16, strict=False, nthreads=8) == 0
    assert np.array_equal(hc_a, o.hitcount.T.astype(np.int64)), "16-spectrum slice: hit counts differ from the oracle"
    assert_close(wf[:16], o.waterfall[:16], "C5 full frame: waterfall rows of the first 16 spectra")
    assert_close(a.waterfall[:16], o.waterfall[:16], "C5 16-spectrum launch: waterfall rows")
    for q in (f, a, b):
        q.close()


def test_c5_bench_launch_shape_vs_oracle(amd, torch_cuda, oracle_built, monkeypatch):
    """The launch shape `bench.py --config C5` times: THREE display frames of 1024 spectra x 65536 points (fp16 IQ, 512 bins, 64 Mi
    samples each) queued back to back -- relaxed input ordering, default streams, NO finish() and no buffer read between them -- so
    that count / scan / merge of frame f run on their own streams in the gaps of frame f + 1's FFT kernel and the intermediate sets
    and hit-count sets are reused across frames in flight (fosphor_api.cpp run(): set / hset rotation).  Semantics: cl.c:870-968
    applied per frame, display.cl:130-178 over each frame's 1024 spectra (the kernel is batch-generic; only the host caps it).
    After the third frame: the last frame's hit counts equal the oracle's bit for bit, the state all three frames went into
    (persistence histogram, live, max-hold, waterfall ring) is in tolerance, and every buffer equals, bit for bit, what the
    single-stream form (FOSPHOR_AMD_OVERLAP=0: one kernel after the other) leaves for the same three frames."""
    torch = torch_cuda
    n, nb, rows, total, frames = 65536, 512, 1024, 1024, 3
    threads = min(os.cpu_count() or 1, 64)
    kw = dict(fft_len_log=16, n_bins=nb, wf_rows=rows, max_spectra=total, max_batches=1, iq_fp16=True)
    g = torch.Generator(device="cuda"); g.manual_seed(1297)
    ds = []
    for k in range(frames):
        d = torch.empty((total * n, 2), dtype=torch.float32, device="cuda").normal_(0.0, [0.05, 0.3, 0.01][k], generator=g)
        if k:
            t = torch.arange(total * n, device="cuda", dtype=torch.float32)
            d[:, 0] += 0.1 * torch.cos(0.173 * k * t); d[:, 1] += 0.1 * torch.sin(0.173 * k * t)
            del t
        ds.append(d.to(torch.float16))
        del d
    torch.cuda.synchronize()
    res = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("FOSPHOR_AMD_OVERLAP", overlap)
        f = amd.Fosphor(**kw)
        f.set_input_ordering(False)
        assert f.finish() >= 0					# boot fills are not part of the shape
        for d in ds:						# all three frames in flight together, as in the timed loop
            assert f.process_device(d, 1, total) == 0
        assert f.finish() >= 0
        res.append((f.hitcount.copy(), f.histogram.copy(), f.waterfall.copy(), f.spectrum.copy(), f.waterfall_pos))
        f.close()
    a, b = res
    assert np.array_equal(a[0], b[0]), "hit counts differ between the three-stream and the single-stream form"
    assert np.array_equal(canon_bits(a[1]), canon_bits(b[1])), "histogram differs between the three-stream and the single-stream form"
    assert np.array_equal(canon_bits(a[2]), canon_bits(b[2])), "waterfall differs between the three-stream and the single-stream form"
    assert np.array_equal(canon_bits(a[3]), canon_bits(b[3])), "spectrum differs between the three-stream and the single-stream form"
    del b, res
    o = Oracle(fft_len_log=16, n_bins=nb, wf_rows=rows)
    for d in ds:
        assert o.process(d.cpu().numpy().astype(np.float32), strict=False, nthreads=threads) == 0
    assert a[4] == o.waterfall_pos
    assert np.array_equal(a[0], o.hitcount.T), "last frame: hit counts differ from the oracle in %d cells" % (a[0] != o.hitcount.T).sum()
    assert np.all(a[0].astype(np.int64).sum(0) == total)
    assert_close(a[2], o.waterfall, "C5 bench shape: waterfall")
    assert_close(a[3][0, :, 1], o.spectrum[0, :, 1], "C5 bench shape: live")
    assert_close(a[3][1, :, 1], o.spectrum[1, :, 1], "C5 bench shape: max-hold")
    assert_hist_close(a[1], o.histogram, "C5 bench shape: histogram after three frames")


def test_c3_bench_launch_shape_vs_oracle(amd, torch_cuda, oracle_built):
    """The launch shape `bench.py --config C3` times, against the oracle: calls of 14 batches of 4096 spectra x 8192 points, 50 %
    overlap fused into the read (overlap_cc_impl.cc:64-79), 512 bins -- 896 tiles of 64 spectra = 4 x 224, so the FFT kernel of a
    call that finds the previous call's count / merge kernels still queued runs on 224 CUs beside them (space sharing, DESIGN.md
    section 8).  TWO such calls back to back, nothing read or waited for in between.  The oracle takes the same 28 batches one
    fosphor_process at a time on the materialised stream: last batch's hit counts bit for bit, the state every batch went into in
    tolerance; and the second call's FFT launch did take the shared form (fosphor_amd_share_stats)."""
    torch = torch_cuda
    n, nb, over, F, B = 8192, 512, 2, 14, 4096
    hop = n // over
    threads = min(os.cpu_count() or 1, 64)
    stream_len = (F * B - 1) * hop + n
    g = torch.Generator(device="cuda"); g.manual_seed(4321)
    ds = []
    for call in range(2):
        d = torch.empty((stream_len, 2), dtype=torch.float32, device="cuda").normal_(0.0, [0.05, 0.4][call], generator=g)
        if call:
            t = torch.arange(stream_len, device="cuda", dtype=torch.float32)
            d[:, 0] += 0.1 * torch.cos(0.21 * t); d[:, 1] += 0.1 * torch.sin(0.21 * t)
            del t
        ds.append(d)
    torch.cuda.synchronize()
    f = amd.Fosphor(fft_len_log=13, n_bins=nb, max_spectra=F * B, max_batches=F)
    f.set_input_ordering(False)
    assert f.finish() >= 0
    for d in ds:							# back to back: the second call's FFT launch shares the chip with the first call's tail
        assert f.process_device_overlap(d, F, B, over) == 0
    assert f.finish() >= 0
    shared, full, cus = f.share_stats()
    assert shared + full == 2 and cus == 224
    assert shared >= 1, "no FFT launch took the space-sharing form: the shape under test did not occur"
    o = Oracle(fft_len_log=13, n_bins=nb)
    idx = (np.arange(B)[:, None] * hop + np.arange(n)[None, :]).reshape(-1)		# overlap_cc: window t starts at sample t * hop
    for d in ds:
        x = d.cpu().numpy()
        for k in range(F):
            xb = x[k * B * hop:(k * B + B - 1) * hop + n]
            assert o.process(xb[idx], strict=False, nthreads=threads) == 0
        del x
    compare_state(f, o, "C3 bench launch shape (2 calls x 14 batches x 4096 spectra, space sharing)")
    f.close()


def test_host_and_pinned_entry_points_interleaved(amd, torch_cuda, oracle_built):
    """fosphor_process (the reference's entry point: host samples, staged through the instance's two pinned slots on its own
    stream) and fosphor_amd_process_pinned / fosphor_amd_upload_pinned (upload on the copy stream, straight from the caller's pinned
    memory) SHARE the two device staging buffers.  Alternating them on one instance must keep the sample stream in order and must
    never let an upload land in a buffer an earlier call's FFT kernel still reads: full 1024-spectrum batches (the widest window for
    such a race), pending uploads overtaken by a fosphor_process call, state against the oracle."""
    import ctypes as C
    torch = torch_cuda
    L = amd.load()
    f = amd.Fosphor(n_bins=256)
    o = Oracle(n_bins=256)
    b = 1024 * 1024
    xs = [add_tone(gaussian_iq(b, 1500 + k, sigma=[0.05, 0.5, 0.005][k % 3]), 0.1, 0.03 * (k + 1)) for k in range(8)]
    pins = [torch.from_numpy(x).pin_memory() for x in xs]
    assert f.process(xs[0]) == 0						# slot 0, H2D on the instance's stream
    assert L.fosphor_amd_process_pinned(f.h, pins[1].data_ptr(), b) == 0	# slot 1, H2D on the copy stream
    assert L.fosphor_amd_process_pinned(f.h, pins[2].data_ptr(), b) == 0	# slot 0 again: must wait for batch 0's FFT kernel
    assert f.process(xs[3]) == 0						# slot 1: must wait for batch 1's
    assert L.fosphor_amd_upload_pinned(f.h, pins[4].data_ptr(), b) == 0	# two uploads pending ...
    assert L.fosphor_amd_upload_pinned(f.h, pins[5].data_ptr(), b) == 0
    assert f.process(xs[6]) == 0						# ... are applied first: the stream stays in order
    assert L.fosphor_amd_pending_uploads(f.h) == 0
    assert L.fosphor_amd_process_pinned(f.h, pins[7].data_ptr(), b) == 0
    for x in xs:
        assert o.process(x, nthreads=8) == 0
    f.draw()
    compare_state(f, o, "fosphor_process and the pinned entry points interleaved (8 batches)")
    f.close()


def _c5_outputs(f):
    return [canon_bits(f.waterfall), canon_bits(f.histogram), canon_bits(f.spectrum), f.hitcount.copy()]


@pytest.mark.parametrize("tile", [None, "4", "16", "nomask"])
def test_c5_call_shapes_vs_oracle(amd, torch_cuda, oracle_built, monkeypatch, tile):
    """N = 65536 (fp16 IQ, 512 bins) through the fused two-level kernel (clusters of 8 work-groups per XCD, the intermediate
    spectrum resident in the XCD's L2) over calls whose tile counts do not divide evenly among the clusters, multi-batch calls
    and a ring wrap, with the tile lengths small launches pick and forced ones (the 9th-bit plane of the bin indices is laid
    out per tile), sparse and dense count hand-off: every call against the oracle."""
    torch = torch_cuda
    n, nb, rows = 65536, 512, 64
    threads = min(os.cpu_count() or 1, 64)
    monkeypatch.delenv("FOSPHOR_AMD_TILE", raising=False)
    monkeypatch.delenv("FOSPHOR_AMD_ROWMASK", raising=False)
    if tile == "nomask":
        tile = None
        monkeypatch.setenv("FOSPHOR_AMD_ROWMASK", "0")
    elif tile:
        monkeypatch.setenv("FOSPHOR_AMD_TILE", tile)
    f = amd.Fosphor(fft_len_log=16, n_bins=nb, wf_rows=rows, max_spectra=160, max_batches=4, iq_fp16=True)
    o = Oracle(fft_len_log=16, n_bins=nb, wf_rows=rows)
    for call, (nbat, batch) in enumerate([(1, 16), (1, 144), (3, 48), (2, 80), (1, 160)]):
        x = add_tone(gaussian_iq(nbat * batch * n, 301 + call), 0.04, 0.11 + 0.05 * call).astype(np.float16)
        d = torch.from_numpy(x).cuda()
        assert f.process_device(d, nbat, batch) == 0
        assert f.finish() >= 0
        x32 = x.astype(np.float32)
        for k in range(nbat):
            assert o.process(x32[k * batch * n:(k + 1) * batch * n], strict=False, nthreads=threads) == 0
        assert f.waterfall_pos == o.waterfall_pos
        assert np.array_equal(f.hitcount, o.hitcount.T), "call %d: hit counts" % call
        assert_close(f.waterfall, o.waterfall, "call %d: waterfall" % call)
        assert_close(f.spectrum[0, :, 1], o.spectrum[0, :, 1], "call %d: live" % call)
        assert_close(f.spectrum[1, :, 1], o.spectrum[1, :, 1], "call %d: max-hold" % call)
        assert_hist_close(f.histogram, o.histogram, "call %d: histogram" % call)
    f.close()


def test_c5_fused_instances_side_by_side(amd, torch_cuda, monkeypatch):
    """Three N = 65536 instances with work queued at the same time on their own streams: the clusters of the fused
    kernels form from whatever work-groups are resident (no kernel waits for a work-group that is not), all three
    finish and agree."""
    torch = torch_cuda
    n, nb, rows = 65536, 512, 64
    fs = [amd.Fosphor(fft_len_log=16, n_bins=nb, wf_rows=rows, max_spectra=128, iq_fp16=True) for _ in range(3)]
    x = add_tone(gaussian_iq(128 * n, 311), 0.05, 0.37).astype(np.float16)
    d = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    for rep in range(3):
        for f in fs:
            assert f.process_device(d, 1, 128) == 0
    for f in fs:
        assert f.finish() >= 0
    ref = _c5_outputs(fs[0])
    for f in fs[1:]:
        for got, want in zip(_c5_outputs(f), ref):
            assert np.array_equal(got, want)
    for f in fs:
        f.close()


@pytest.mark.parametrize("n_bins,wf_rows,consts", [
    (16, 1024, None), (64, 256, None), (192, 1024, (4.0, 256.0, 0.01)), (256, 2048, (32.0, 4096.0, 0.0005)),
])
def test_geometry_and_constant_variants(amd, torch_cuda, oracle_built, n_bins, wf_rows, consts):
    """Bins 16..256, waterfall depth, rise/decay/averaging constants (cl.c:714-716) other than the
    reference's: oracle-defined generalisations, same parity bars."""
    kw = {}
    if consts:
        kw = dict(t0r=consts[0], t0d=consts[1], alpha=consts[2])
    f = amd.Fosphor(n_bins=n_bins, wf_rows=wf_rows, max_spectra=2048, **kw)
    o = Oracle(n_bins=n_bins, wf_rows=wf_rows)
    if consts:
        o.set_constants(*consts)
    f.set_power_range(-10, 5)
    o.set_power_range(-10, 5)
    import torch
    for k, b in enumerate([48, 1024, 2048, 16]):
        x = add_tone(gaussian_iq(b * 1024, 200 + k), 0.02, 0.4, t0=k * 7)
        if b <= 1024:
            assert f.process(x) == 0
        else:			# beyond the host-side cap (cl.c:885): device path, one display launch
            assert f.process(x) == -errno.EINVAL
            assert f.process_device(torch.from_numpy(x).cuda(), 1, b) == 0
        assert o.process(x, strict=False, nthreads=8) == 0
        compare_state(f, o, "bins %d rows %d call %d" % (n_bins, wf_rows, k))
    f.close()


def test_capacity_and_argument_errors(amd, torch_cuda):
    torch = torch_cuda
    f = amd.Fosphor(max_spectra=64, max_batches=2)
    d = torch.zeros((64 * 1024, 2), dtype=torch.float32, device="cuda")
    assert f.process_device(d, 1, 64) == 0
    assert f.process_device(d, 1, 80) == -errno.EINVAL		# over capacity
    assert f.process_device(d, 4, 16) == -errno.EINVAL		# more batches than max_batches
    assert f.process_device(d, 1, 24) == -errno.EINVAL		# not a multiple of 16 spectra
    assert f.process_device(d, 0, 16) == -errno.EINVAL
    assert f.accumulate_device(d, 32, 8, 64) == -errno.EINVAL	# shard offset not on a 16-spectrum boundary
    assert f.accumulate_device(d, 32, 48, 64) == -errno.EINVAL	# shard runs past the batch
    with pytest.raises(RuntimeError):
        amd.Fosphor(n_bins=100)					# not a multiple of 16
    with pytest.raises(RuntimeError):
        amd.Fosphor(wf_rows=1000)				# not a power of two
    with pytest.raises(RuntimeError):
        amd.Fosphor(fft_len_log=11)				# only 2^10, 2^13 and 2^16
    # fosphor_amd_share_stats: zeros for an instance that never shares (N = 1024); the split this device would use; NULL pointers allowed
    import ctypes as C
    L = amd.load()
    assert f.share_stats() == (0, 0, 224)
    assert L.fosphor_amd_share_stats(f.h, None, None, None) == 0
    assert L.fosphor_amd_share_stats(None, None, None, None) == -errno.EINVAL
    f.close()
