"""CPU: pin the oracle (oracle/fosphor_oracle.c) against fixtures produced by the
REFERENCE's own kernels (tests/golden/*.npz, made by oracle/gen_golden.py).

Bit-exact everywhere: the restatement mirrors the reference's operation order and uses the
same built-in binding the fixtures were generated with.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_cases as gc
from oracle_lib import Oracle, RefKernels, canon_bits, digest, have_ref, hitcount_from_rows

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLD, "golden_meta.json")))


def sha(a):
    return digest(a)


def bits(a):
    return canon_bits(a)


def make_oracle(spec):
    o = Oracle()
    if "power_range" in spec:
        o.set_power_range(*spec["power_range"])
    if "window" in spec:
        o.set_window(spec["window"]())
    return o


@pytest.mark.parametrize("name", list(gc.CASES))
def test_restatement_matches_reference_fixture(name, oracle_built):
    spec = gc.CASES[name]
    if name == "c8_b8192" and os.environ.get("FOSPHOR_SKIP_SLOW"):
        pytest.skip("slow")
    z = np.load(os.path.join(GOLD, name + ".npz"))
    o = make_oracle(spec)
    calls = spec["calls"]()
    for k, x in enumerate(calls):
        m = META[name]["calls"][k]
        pre = "c%d_" % k
        if spec["store"] == "full":
            assert np.array_equal(bits(x), bits(z[pre + "x"])), "input recipe drifted from fixture"
        assert o.waterfall_pos == m["pos0"]
        rv = o.process(x, strict=spec.get("strict", True), nthreads=4)
        assert rv == 0
        assert o.waterfall_pos == m["pos1"]
        assert abs(o.histo_scale - m["hs"]) == 0 and abs(o.histo_offset - m["ho"]) == 0
        # digests of every output, every call
        assert sha(o.fft_out) == m["sha_fft"]
        assert sha(o.waterfall) == m["sha_wf"]
        assert sha(o.histogram) == m["sha_hist"]
        assert sha(o.spectrum) == m["sha_spec"]
        if "sha_hc" in m:
            assert sha(o.hitcount) == m["sha_hc"]			# integer counts, bit-exact
            assert int(o.hitcount.sum()) == m["batch"] * 1024
        if spec["store"] == "full":
            assert np.array_equal(bits(o.fft_out), bits(z[pre + "fft"]))
            assert np.array_equal(bits(o.waterfall[z[pre + "wf_idx"]]), bits(z[pre + "wf_rows"]))
            assert np.array_equal(bits(o.histogram), bits(z[pre + "hist"]))
            assert np.array_equal(bits(o.spectrum), bits(z[pre + "spec"]))
            assert np.array_equal(o.hitcount, z[pre + "hc"])
            # the fixture's counts are themselves derived from the reference's pwr rows
            assert np.array_equal(hitcount_from_rows(z[pre + "wf_rows"], m["hs"], m["ho"], 128), z[pre + "hc"])
        elif k == len(calls) - 1:
            assert np.array_equal(bits(o.histogram), bits(z[pre + "hist"]))
            assert np.array_equal(bits(o.spectrum), bits(z[pre + "spec"]))


def test_fft512_known_answer(oracle_built):
    z = np.load(os.path.join(GOLD, "fft512.npz"))
    y = Oracle.fft(z["x"], z["win"], fft_len_log=9)
    assert np.array_equal(bits(y), bits(z["fft"]))
    assert sha(y) == META["fft512"]["sha_fft"]
    # and it is a DFT: forward, unnormalised, natural order (fft.cl:357-394)
    x = z["x"].reshape(-1, 512, 2).astype(np.float64)
    ref = np.fft.fft((x[..., 0] + 1j * x[..., 1]) * z["win"].astype(np.float64), axis=-1)
    got = y[..., 0].astype(np.float64) + 1j * y[..., 1]
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 5e-7


def test_fft1024_is_a_dft(oracle_built):
    z = np.load(os.path.join(GOLD, "c1_gauss_b16.npz"))
    o = Oracle()
    x = z["c0_x"].reshape(16, 1024, 2).astype(np.float64)
    ref = np.fft.fft((x[..., 0] + 1j * x[..., 1]) * o.window.astype(np.float64), axis=-1)
    got = z["c0_fft"][..., 0].astype(np.float64) + 1j * z["c0_fft"][..., 1]
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 5e-7


@pytest.mark.parametrize("log2n", [13, 16])
def test_long_fft_plans_are_dfts(oracle_built, log2n):
    """No reference behaviour exists beyond N = 1024: the plans are the oracle's -- the reference's Stockham data flow (8192: radix
    16.16.16.2, fft.cl:278-350,397-466 generalised; 65536: radix 16.16.16.16) with the arithmetic inside a pass restated as
    fused-multiply-add butterflies that carry their twiddles (o_pass_radix16_fma / o_pass_radix2_fma, round 5).
    Whatever the plan, it must BE a forward, unnormalised, natural-order DFT -- as accurate against numpy's fp64 FFT as the
    reference's own 1024-point kernel is."""
    n = 1 << log2n
    rng = np.random.default_rng(1300 + log2n)
    x = (rng.standard_normal((2 * n, 2)) * 0.5).astype(np.float32)
    win = (0.5 + 0.5 * rng.random(n)).astype(np.float32)
    y = Oracle.fft(x, win, fft_len_log=log2n)
    xr = x.reshape(2, n, 2).astype(np.float64)
    ref = np.fft.fft((xr[..., 0] + 1j * xr[..., 1]) * win.astype(np.float64), axis=-1)
    got = y.reshape(2, n, 2)[..., 0].astype(np.float64) + 1j * y.reshape(2, n, 2)[..., 1]
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-6


def test_edge_semantics_in_fixtures():
    """Properties the reference fixtures exhibit (documented behaviour the HIP path must keep)."""
    z = np.load(os.path.join(GOLD, "c3_zero_b16.npz"))
    assert np.all(np.isneginf(z["c0_wf_rows"]))			# log10(hypot(0,0)) = -inf, display.cl:136
    assert np.all(z["c0_hc"][:, 0] == 16) and z["c0_hc"][:, 1:].sum() == 0	# -> bin 0
    z = np.load(os.path.join(GOLD, "c4_fullscale_b16.npz"))
    assert z["c0_hc"][256, 127] == 16					# tone column clamps to the top bin
    z = np.load(os.path.join(GOLD, "c9_nonfinite_b16x2.npz"))
    hc = z["c0_hc"]
    assert np.all(hc.sum(1) == 16)
    assert np.all(hc[:, 0] >= 3)					# NaN, inf and zero spectra all land in bin 0
    # live spectrum recovers from a non-finite state on the next call (display.cl:206-207)
    assert not np.all(np.isfinite(z["c0_spec"][0, :, 1]))
    assert np.all(np.isfinite(z["c1_spec"][0, :, 1]))


def test_process_argument_errors(oracle_built):
    """cl.c:882-886: len must be a multiple of 16*1024 and at most 1024*1024."""
    import errno
    o = Oracle()
    x = np.zeros((17 * 1024, 2), np.float32)
    assert o.process(x) == -errno.EINVAL
    x = np.zeros((1040 * 1024, 2), np.float32)
    assert o.process(x, strict=True) == -errno.EINVAL
    assert o.waterfall_pos == 0


def test_restatement_threads_agree(oracle_built):
    x = gc.CASES["c2_tone_b32x3"]["calls"]()[0]
    a, b = Oracle(), Oracle()
    a.process(x, nthreads=1)
    b.process(x, nthreads=8)
    for f in ("waterfall", "histogram", "spectrum", "hitcount", "fft_out"):
        assert digest(getattr(a, f)) == digest(getattr(b, f))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_kernels_live_crosscheck(oracle_built):
    """Where the compiled reference kernels exist, run them again on a fresh random input
    (not a stored fixture) against the restatement: bit-exact."""
    rng = np.random.default_rng(12345)
    x = (rng.standard_normal((48 * 1024, 2)) * 0.3).astype(np.float32)
    r, o = RefKernels(portable=True), Oracle()
    for blk in (x[:16 * 1024], x[16 * 1024:]):
        assert r.process(blk) == 0 and o.process(blk) == 0
        for f in ("fft_out", "waterfall", "histogram", "spectrum"):
            assert np.array_equal(bits(getattr(r, f)), bits(getattr(o, f))), f


GLIBC_CELL_BUDGET_PER_MI = 8


@pytest.mark.parametrize("name", ["c1_gauss_b16", "c2_tone_b32x3", "c5_wrap_b512_b1024", "c6_range_m20_5"])
def test_informational_oracle_vs_glibc_bound_reference(oracle_built, name):
    """INFORMATIONAL, with a stated budget: the restatement (pinned portable math) against the reference kernels
    run with glibc's sinf/cosf/hypotf/log10f/roundf (tests/golden/glibc_binding_hc.npz).  The OpenCL built-ins are
    implementation-defined, so neither binding is 'the' reference; the two differ in at most
    GLIBC_CELL_BUDGET_PER_MI hit-count cells per 2^20 samples (measured 4 in the 1024-spectrum call)."""
    zg = np.load(os.path.join(GOLD, "glibc_binding_hc.npz"))
    spec = gc.CASES[name]
    o = Oracle()
    if "power_range" in spec:
        o.set_power_range(*spec["power_range"])
    for k, x in enumerate(spec["calls"]()):
        assert o.process(x, nthreads=8) == 0
        hc = o.hitcount.astype(np.int64)
        ref = zg["%s_c%d_hc" % (name, k)].astype(np.int64)
        differ = int((hc != ref).sum())
        budget = max(GLIBC_CELL_BUDGET_PER_MI, GLIBC_CELL_BUDGET_PER_MI * x.shape[0] // (1 << 20))
        assert differ <= budget, "%s call %d: %d cells differ (budget %d)" % (name, k, differ, budget)
