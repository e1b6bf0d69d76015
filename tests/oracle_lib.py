"""ctypes bindings for the parity checkers under oracle/ (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (gr-fosphor_amd/) never does.

  Oracle     -- oracle/libfosphor_oracle.so, the plain-C restatement of
                lib/fosphor/{fft.cl,display.cl,cl.c,fosphor.c}; travels everywhere.
  RefKernels -- oracle/_ref/libfosphor_ref.so, the reference's own kernel sources
                compiled for x86 (oracle/Makefile `make ref`); exists only where it
                was built from /root/reference (the .so travels to the GPU box, the
                sources do not).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libfosphor_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libfosphor_ref.so")

_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def build_oracle(ref=False):
    """Compile the checkers (gcc; plus clang for the reference kernels when asked)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR] + targets)


def have_ref():
    return os.path.exists(REF_SO)


def _as_f32(x):
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a


class Oracle:
    """CPU restatement.  Geometry defaults are the reference's (private.h:21-25)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                build_oracle()
            L = C.CDLL(ORACLE_SO)
            L.fosphor_oracle_new.restype = C.c_void_p
            L.fosphor_oracle_new.argtypes = [C.c_int, C.c_int, C.c_int]
            L.fosphor_oracle_free.argtypes = [C.c_void_p]
            L.fosphor_oracle_set_window_default.argtypes = [C.c_void_p]
            L.fosphor_oracle_set_window.argtypes = [C.c_void_p, C.c_void_p]
            L.fosphor_oracle_set_power_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.fosphor_oracle_set_constants.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
            L.fosphor_oracle_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
            L.fosphor_oracle_fft.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            L.fosphor_oracle_bin.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
            L.fosphor_oracle_twiddle.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            L.fosphor_oracle_bins.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
            for n in ("waterfall", "histogram", "spectrum", "fft_out"):
                f = getattr(L, "fosphor_oracle_" + n)
                f.restype = _fp
                f.argtypes = [C.c_void_p]
            L.fosphor_oracle_window.restype = _fp
            L.fosphor_oracle_window.argtypes = [C.c_void_p]
            L.fosphor_oracle_hitcount.restype = _u32p
            L.fosphor_oracle_hitcount.argtypes = [C.c_void_p]
            L.fosphor_oracle_waterfall_pos.argtypes = [C.c_void_p]
            L.fosphor_oracle_histo_scale.restype = C.c_float
            L.fosphor_oracle_histo_scale.argtypes = [C.c_void_p]
            L.fosphor_oracle_histo_offset.restype = C.c_float
            L.fosphor_oracle_histo_offset.argtypes = [C.c_void_p]
            cls._lib = L
        return cls._lib

    def __init__(self, fft_len_log=10, n_bins=128, wf_rows=1024):
        self.L = self.lib()
        self.log2n, self.n, self.n_bins, self.wf_rows = fft_len_log, 1 << fft_len_log, n_bins, wf_rows
        self.h = self.L.fosphor_oracle_new(fft_len_log, n_bins, wf_rows)
        self.last_batch = 0

    def __del__(self):
        try:
            self.L.fosphor_oracle_free(self.h)
        except Exception:
            pass

    def set_window(self, win):
        w = _as_f32(win)
        assert w.size == self.n
        self.L.fosphor_oracle_set_window(self.h, w.ctypes.data)

    def set_window_default(self):
        self.L.fosphor_oracle_set_window_default(self.h)

    def set_power_range(self, db_ref, db_per_div):
        self.L.fosphor_oracle_set_power_range(self.h, db_ref, db_per_div)

    def set_constants(self, t0r, t0d, alpha):
        self.L.fosphor_oracle_set_constants(self.h, t0r, t0d, alpha)

    def process(self, samples, strict=True, nthreads=1):
        """samples: float32 array of interleaved (re, im); returns the reference's int code."""
        x = _as_f32(samples).reshape(-1)
        n = x.size // 2
        rv = self.L.fosphor_oracle_process(self.h, x.ctypes.data, n, 1 if strict else 0, nthreads)
        if rv == 0:
            self.last_batch = n // self.n
        return rv

    def _get(self, name, shape):
        p = getattr(self.L, "fosphor_oracle_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=shape).copy()

    @property
    def window(self):
        return self._get("window", (self.n,))

    @property
    def waterfall(self):
        return self._get("waterfall", (self.wf_rows, self.n))

    @property
    def histogram(self):
        return self._get("histogram", (self.n_bins, self.n))

    @property
    def spectrum(self):
        """[2][N][2]: live then max-hold, (x, y) vertices, fft-shifted index."""
        return self._get("spectrum", (2, self.n, 2))

    @property
    def hitcount(self):
        return self._get("hitcount", (self.n, self.n_bins))

    @property
    def fft_out(self):
        return self._get("fft_out", (self.last_batch, self.n, 2))

    @property
    def waterfall_pos(self):
        return self.L.fosphor_oracle_waterfall_pos(self.h)

    @property
    def histo_scale(self):
        return self.L.fosphor_oracle_histo_scale(self.h)

    @property
    def histo_offset(self):
        return self.L.fosphor_oracle_histo_offset(self.h)

    @classmethod
    def fft(cls, x, win, fft_len_log=10):
        L = cls.lib()
        n = 1 << fft_len_log
        x = _as_f32(x).reshape(-1, n, 2)
        w = _as_f32(win)
        out = np.empty_like(x)
        L.fosphor_oracle_fft(fft_len_log, x.ctypes.data, out.ctypes.data, w.ctypes.data, x.shape[0])
        return out

    @classmethod
    def bin(cls, re, im, hs, ho, n_bins=128):
        return cls.lib().fosphor_oracle_bin(re, im, hs, ho, n_bins)


def oracle_bins(fft, hs, ho, n_bins=128):
    """(bin int32[n], pwr float32[n]) of FFT outputs float32[n][2] through the pinned pipeline."""
    L = Oracle.lib()
    f = _as_f32(fft).reshape(-1, 2)
    b = np.empty(f.shape[0], np.int32)
    p = np.empty(f.shape[0], np.float32)
    L.fosphor_oracle_bins(f.ctypes.data, f.shape[0], hs, ho, n_bins, b.ctypes.data, p.ctypes.data)
    return b, p


class RefKernels:
    """The reference's fft.cl / display.cl executed on the host (N=1024, 128 bins only)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(REF_SO)
            L.ref_new.restype = C.c_void_p
            L.ref_free.argtypes = [C.c_void_p]
            L.ref_set_binding.argtypes = [C.c_int]
            L.ref_set_window.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_set_power_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.ref_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            L.ref_fft.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            for n in ("waterfall", "histogram", "spectrum", "fft_out"):
                f = getattr(L, "ref_" + n)
                f.restype = _fp
                f.argtypes = [C.c_void_p]
            L.ref_waterfall_pos.argtypes = [C.c_void_p]
            L.ref_histo_scale.restype = C.c_float
            L.ref_histo_scale.argtypes = [C.c_void_p]
            L.ref_histo_offset.restype = C.c_float
            L.ref_histo_offset.argtypes = [C.c_void_p]
            cls._lib = L
        return cls._lib

    def __init__(self, portable=True):
        self.L = self.lib()
        self.L.ref_set_binding(1 if portable else 0)
        self.portable = portable
        self.n, self.n_bins, self.wf_rows = 1024, 128, 1024
        self.h = self.L.ref_new()
        self.last_batch = 0

    def __del__(self):
        try:
            self.L.ref_free(self.h)
        except Exception:
            pass

    def set_window(self, win):
        w = _as_f32(win)
        self.L.ref_set_window(self.h, w.ctypes.data)

    def set_power_range(self, db_ref, db_per_div):
        self.L.ref_set_power_range(self.h, db_ref, db_per_div)

    def process(self, samples, strict=True):
        self.L.ref_set_binding(1 if self.portable else 0)
        x = _as_f32(samples).reshape(-1)
        n = x.size // 2
        rv = self.L.ref_process(self.h, x.ctypes.data, n, 1 if strict else 0)
        if rv == 0:
            self.last_batch = n // 1024
        return rv

    def _get(self, name, shape):
        return np.ctypeslib.as_array(getattr(self.L, "ref_" + name)(self.h), shape=shape).copy()

    waterfall = property(lambda s: s._get("waterfall", (1024, 1024)))
    histogram = property(lambda s: s._get("histogram", (128, 1024)))
    spectrum = property(lambda s: s._get("spectrum", (2, 1024, 2)))
    fft_out = property(lambda s: s._get("fft_out", (s.last_batch, 1024, 2)))
    waterfall_pos = property(lambda s: s.L.ref_waterfall_pos(s.h))
    histo_scale = property(lambda s: s.L.ref_histo_scale(s.h))
    histo_offset = property(lambda s: s.L.ref_histo_offset(s.h))

    @classmethod
    def fft(cls, x, win, n=1024, portable=True):
        L = cls.lib()
        L.ref_set_binding(1 if portable else 0)
        x = _as_f32(x).reshape(-1, n, 2)
        w = _as_f32(win)
        out = np.empty_like(x)
        L.ref_fft(n, x.ctypes.data, out.ctypes.data, w.ctypes.data, x.shape[0])
        return out


def canon_bits(a):
    """uint32 view of a float32 array with every NaN mapped to one quiet-NaN pattern.
    NaN sign/payload depends on SSE operand order chosen by the compiler, not on the algorithm."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).copy()
    u[np.isnan(a)] = 0x7FC00000
    return u


def digest(a):
    """sha256 over canonical bits (floats) or raw bytes (integers)."""
    import hashlib
    a = np.asarray(a)
    if a.dtype == np.float32:
        a = canon_bits(a)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------
# Synthetic inputs shared by the fixture generator, the tests and the bench
# (white complex Gaussian, sigma 0.05 per component, numpy default_rng)
# ---------------------------------------------------------------------------

def gaussian_iq(n_samples, seed, sigma=0.05):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n_samples, 2)) * sigma).astype(np.float32)


def add_tone(x, amp, freq, phase0=0.0, t0=0):
    t = np.arange(t0, t0 + x.shape[0], dtype=np.float64)
    ph = 2.0 * np.pi * freq * t + phase0
    y = x.copy()
    y[:, 0] += (amp * np.cos(ph)).astype(np.float32)
    y[:, 1] += (amp * np.sin(ph)).astype(np.float32)
    return y


def hitcount_from_rows(pwr_rows, hs, ho, n_bins):
    """Integer hit counts from the exact pwr rows (the waterfall texels ARE the pwr values
    used for binning: display.cl:136,146,161).  numpy float32 arithmetic = IEEE, same as C."""
    p = np.asarray(pwr_rows, dtype=np.float32)
    v = np.float32(hs) * (p + np.float32(ho))
    finite = np.isfinite(v)
    a = np.abs(np.where(finite, v, 0)).astype(np.float32)
    t = np.trunc(a)
    r = np.where((a - t) >= np.float32(0.5), t + 1, t)
    r = np.copysign(r, np.where(finite, v, 0))
    b = np.clip(r, 0, n_bins - 1).astype(np.int64)
    b = np.where(finite, b, 0)
    n = p.shape[1]
    hc = np.zeros((n, n_bins), dtype=np.uint32)
    for x in range(n):
        hc[x] = np.bincount(b[:, x], minlength=n_bins).astype(np.uint32)
    return hc
