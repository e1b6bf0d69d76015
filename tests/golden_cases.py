"""Recipes for the golden parity cases (SURVEY 8c list).

Shared by oracle/gen_golden.py (which runs the REFERENCE kernels, oracle/_ref, on them and
writes tests/golden/*.npz), by tests/test_oracle_golden.py (restatement vs fixtures) and by
the GPU parity tests (HIP path vs restatement / fixtures).

Every case is reference geometry: N=1024, 128 bins, 1024 waterfall rows.
A case = optional setup + a list of process() calls; inputs are deterministic functions of
numpy default_rng seeds, and are ALSO stored in the small fixtures.
"""
import numpy as np

from oracle_lib import add_tone, gaussian_iq

N = 1024


def blackman_harris(n=N):
    """4-term Blackman-Harris, un-normalised (the GR sink's default window family,
    base_sink_c_impl.cc:55,251-255; the exact GNU Radio array is an INPUT we do not pin)."""
    k = np.arange(n, dtype=np.float64)
    w = (0.35875 - 0.48829 * np.cos(2 * np.pi * k / n) + 0.14128 * np.cos(4 * np.pi * k / n)
         - 0.01168 * np.cos(6 * np.pi * k / n))
    return w.astype(np.float32)


def hann512():
    k = np.arange(512, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2 * np.pi * k / 512)).astype(np.float32)


def _c1():
    return [gaussian_iq(16 * N, 1)]


def _c2():
    calls = []
    for c in range(3):
        x = gaussian_iq(32 * N, 20 + c)
        calls.append(add_tone(x, 0.2, 0.123, t0=c * 32 * N))
    return calls


def _c3():
    return [np.zeros((16 * N, 2), np.float32)]


def _c4():
    return [add_tone(gaussian_iq(16 * N, 4), 40.0, 0.25)]


def _c5():
    return [gaussian_iq(512 * N, 5), gaussian_iq(1024 * N, 6)]


def _c6():
    return [gaussian_iq(16 * N, 7)]


def _c8():
    return [gaussian_iq(8192 * N, 8)]


def _c9():
    x = gaussian_iq(16 * N, 9).reshape(16, N, 2)
    x[3, 5, 0] = np.nan
    x[7, 100, 1] = np.inf
    x[9] *= np.float32(1e30)
    x[11] *= np.float32(1e-30)
    x[12] *= np.float32(1e-42)		# denormal inputs
    x[14] = 0.0
    return [x.reshape(-1, 2), gaussian_iq(16 * N, 10)]


def _c10():
    return [add_tone(gaussian_iq(16 * N, 11, sigma=0.01), 0.5, -0.3)]


# name -> dict(setup, calls, strict, store)
#   store = "full":   inputs + fft + written waterfall rows + histogram + spectrum per call
#   store = "digest": sha256 of every output per call + histogram/spectrum of the last call
CASES = {
    "c1_gauss_b16":      dict(calls=_c1, store="full"),
    "c2_tone_b32x3":     dict(calls=_c2, store="full"),
    "c3_zero_b16":       dict(calls=_c3, store="full"),
    "c4_fullscale_b16":  dict(calls=_c4, store="full"),
    "c5_wrap_b512_b1024": dict(calls=_c5, store="digest"),
    "c6_range_m20_5":    dict(calls=_c6, store="full", power_range=(-20, 5)),
    "c8_b8192":          dict(calls=_c8, store="digest", strict=False),
    "c9_nonfinite_b16x2": dict(calls=_c9, store="full"),
    "c10_blackman_b16":  dict(calls=_c10, store="full", window=blackman_harris),
}

# FFT-only known-answer case for the dormant 512-point kernel (fft.cl:357-394)
FFT512_SEED = 512


def fft512_input():
    return gaussian_iq(4 * 512, FFT512_SEED, sigma=1.0)


# Frequency-axis cases for the label formatter (axis.c:63-162): (center Hz, span Hz, n_div)
AXIS_CASES = [
    (100e6, 2e6, 10), (0.0, 1.0, 10), (0.0, 2e6, 10), (100e6, 0.0, 10), (2.4e9, 20e6, 10), (1e3, 100.0, 10),
    (433.92e6, 250e3, 10), (10.7e6, 48e3, 8), (-5e6, 1e6, 10), (1.57542e9, 4.092e6, 10), (145.8e6, 12.5e3, 10),
    (7.1e6, 192e3, 12), (28.074e6, 3e3, 6), (5.8e9, 160e6, 10), (1.0, 0.5, 10), (99999.0, 10.0, 10),
    (1e12, 1e9, 10), (433.92e6, 250e3, 7), (0.0, 56e6, 14), (88.3e6, 3.2e6, 16),
]
