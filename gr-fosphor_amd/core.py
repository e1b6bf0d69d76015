"""Host-side mirror of the reference's libfosphor interface over the HIP library.

Same names and argument meaning as lib/fosphor/fosphor.h (process / draw / set_fft_window /
set_power_range / set_frequency_range) plus the plain-buffer accessors that replace the
CL<->GL interop.  Everything numeric happens in libfosphor_amd.so on the GPU; this class
only marshals pointers.
"""
import ctypes as C
import errno

import numpy as np

from . import _lib

FFT_LEN_LOG = 10		# private.h:21
FFT_LEN = 1 << FFT_LEN_LOG
MULT_BATCH = 16			# private.h:24
MAX_BATCH = 1024		# private.h:25


def _ptr(x):
    """Device pointer of a torch tensor / anything with data_ptr(), or a raw integer."""
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class Fosphor:
    """One fosphor instance (struct fosphor).  Reference geometry by default."""

    def __init__(self, n_bins=128, wf_rows=1024, fft_len_log=FFT_LEN_LOG, t0r=0.0, t0d=0.0, alpha=0.0,
                 device=-1, max_spectra=1024, max_batches=0, stream=None, iq_fp16=False):
        self.L = _lib.load()
        cfg = _lib.Config(fft_len_log, n_bins, wf_rows, t0r, t0d, alpha, device, max_spectra, max_batches,
                          C.c_void_p(stream) if stream else None, 1 if iq_fp16 else 0)
        self.iq_fp16 = bool(iq_fp16)
        self.h = self.L.fosphor_amd_init(C.byref(cfg))
        if not self.h:
            raise RuntimeError("fosphor_amd_init failed (see stderr); no CPU fallback exists")
        self.n, self.n_bins, self.wf_rows = 1 << fft_len_log, n_bins, wf_rows
        self.max_spectra = max_spectra

    def close(self):
        if getattr(self, "h", None):
            self.L.fosphor_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API ---------------------------------------------------
    def process(self, samples):
        """fosphor_process: host interleaved fp32 (re, im); returns 0 / -EINVAL / -EIO."""
        x = np.ascontiguousarray(samples, dtype=np.float16 if self.iq_fp16 else np.float32).reshape(-1)
        return self.L.fosphor_process(self.h, x.ctypes.data, x.size // 2)

    def draw(self, render=None):
        r = render if render is not None else _lib.Render()
        self.L.fosphor_draw(self.h, C.byref(r))
        return r._wf_pos

    def set_fft_window_default(self):
        self.L.fosphor_set_fft_window_default(self.h)

    def set_fft_window(self, win):
        w = np.ascontiguousarray(win, dtype=np.float32)
        if w.size != self.n:
            raise ValueError("window must have %d taps" % self.n)
        self.L.fosphor_set_fft_window(self.h, w.ctypes.data)

    def set_power_range(self, db_ref, db_per_div):
        self.L.fosphor_set_power_range(self.h, int(db_ref), int(db_per_div))

    def set_frequency_range(self, center, span):
        self.L.fosphor_set_frequency_range(self.h, float(center), float(span))

    # ---- device-resident data path -----------------------------------------
    def process_device(self, d_samples, n_batches, batch):
        return self.L.fosphor_amd_process_device(self.h, _ptr(d_samples), int(n_batches), int(batch))

    def process_device_overlap(self, d_samples, n_batches, batch, overlap):
        """overlap_cc(wlen=N, overlap) fused into the read (unexpanded stream in HBM)."""
        return self.L.fosphor_amd_process_device_overlap(self.h, _ptr(d_samples), int(n_batches), int(batch), int(overlap))

    def finish(self):
        return self.L.fosphor_amd_finish(self.h)

    def accumulate_device(self, d_samples, n_local, t_offset, total_batch, overlap=1):
        if overlap > 1:
            return self.L.fosphor_amd_accumulate_device_overlap(self.h, _ptr(d_samples), n_local, t_offset, total_batch, overlap)
        return self.L.fosphor_amd_accumulate_device(self.h, _ptr(d_samples), n_local, t_offset, total_batch)

    def merge(self, total_batch):
        return self.L.fosphor_amd_merge(self.h, total_batch)

    def set_partial_slot(self, slot):
        return self.L.fosphor_amd_set_partial_slot(self.h, slot)

    def partials(self):
        p = _lib.Partials()
        self.L.fosphor_amd_get_partials(self.h, C.byref(p))
        return p

    # ---- native exchange (RCCL, include/fosphor_amd.h) ------------------------
    def exchange(self, comm):
        return self.L.fosphor_amd_exchange(self.h, comm)

    def exchange_sliced(self, comm, world, rank):
        return self.L.fosphor_amd_exchange_sliced(self.h, comm, world, rank)

    def merge_sliced(self, total_batch, world, rank):
        return self.L.fosphor_amd_merge_sliced(self.h, total_batch, world, rank)

    def gather_state(self, comm, world, rank):
        return self.L.fosphor_amd_gather_state(self.h, comm, world, rank)

    def buffers(self, hitcount=True):
        """struct fosphor_amd_buffers; hitcount=False: no export kernel, no wait, d_hitcount is NULL."""
        b = _lib.Buffers()
        rv = (self.L.fosphor_amd_get_buffers if hitcount else self.L.fosphor_amd_get_buffers_nohc)(self.h, C.byref(b))
        if rv:
            raise RuntimeError("fosphor_amd_get_buffers -> %d" % rv)
        return b

    # ---- results as host arrays ------------------------------------------
    def _read(self, which, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        rv = self.L.fosphor_amd_read(self.h, which, out.ctypes.data, out.nbytes)
        if rv:
            raise RuntimeError("fosphor_amd_read(%d) -> %d (%s)" % (which, rv, errno.errorcode.get(-rv, "?")))
        return out

    @property
    def waterfall(self):
        return self._read(0, (self.wf_rows, self.n), np.float32)

    @property
    def histogram(self):
        return self._read(1, (self.n_bins, self.n), np.float32)

    @property
    def spectrum(self):
        return self._read(2, (2, self.n, 2), np.float32)

    @property
    def hitcount(self):
        """uint32 [n_bins][N] of the last batch (the oracle's view is the transpose)."""
        return self._read(3, (self.n_bins, self.n), np.uint32)

    @property
    def waterfall_pos(self):
        return self.buffers(False).waterfall_pos

    def colorize(self, image, palette=None, scale=None, offset=None, rows=None):
        """RGBA8 picture of the waterfall (image=0, newest row first) or the histogram (image=1,
        highest bin first), fft-shifted, as a torch uint8 tensor [rows][N][4] on the device.
        palette: numpy uint32[n] (host) or None for the reference's 256-entry palette of that image;
        scale/offset None: the reference's values (include/fosphor_amd_cmap.h)."""
        import torch
        if rows is None:
            rows = self.wf_rows if image == 0 else self.n_bins
        out = torch.empty((rows, self.n), dtype=torch.int32, device="cuda")
        if palette is not None:
            palette = np.ascontiguousarray(palette, dtype=np.uint32)
        defaults = scale is None and offset is None
        rv = self.L.fosphor_amd_colorize(self.h, int(image), palette.ctypes.data if palette is not None else None,
                                         palette.size if palette is not None else 0, 1 if defaults else 0,
                                         float(scale or 0.0), float(offset or 0.0), int(rows), out.data_ptr())
        if rv:
            raise RuntimeError("fosphor_amd_colorize -> %d" % rv)
        return out.view(torch.uint8).reshape(rows, self.n, 4)

    @property
    def histo_scale(self):
        return self.buffers(False).histo_scale

    @property
    def histo_offset(self):
        return self.buffers(False).histo_offset

    # ---- kernel-level hooks -------------------------------------------------
    def fft_device(self, d_in, d_out, n_spectra):
        return self.L.fosphor_amd_fft(self.h, _ptr(d_in), _ptr(d_out), n_spectra)

    def bin_device(self, d_fft, d_bin, d_pwr, n):
        return self.L.fosphor_amd_bin(self.h, _ptr(d_fft), _ptr(d_bin), _ptr(d_pwr), n)

    # ---- measurement --------------------------------------------------------
    def profile(self, enable=True):
        """False/0: off; True/1: hipEvents around every kernel; 2: around K1 only."""
        self.L.fosphor_amd_profile(self.h, 2 if enable == 2 and enable is not True else (1 if enable else 0))

    def tune_placement(self, d_samples, n_batches, batch, max_tries=6):
        """fosphor_amd_tune_placement: (re-allocations made, slowest set's twin us before, after)."""
        b, a = C.c_float(), C.c_float()
        rv = self.L.fosphor_amd_tune_placement(self.h, _ptr(d_samples), int(n_batches), int(batch), int(max_tries), C.byref(b), C.byref(a))
        if rv < 0:
            raise RuntimeError("fosphor_amd_tune_placement -> %d" % rv)
        return rv, b.value, a.value

    def traffic_twin(self, d_samples, n_batches, batch, reps=20):
        """ms per launch of K1's memory traffic alone (include/fosphor_amd.h)"""
        ms = C.c_float()
        rv = self.L.fosphor_amd_traffic_twin(self.h, _ptr(d_samples), int(n_batches), int(batch), int(reps), C.byref(ms))
        if rv:
            raise RuntimeError("fosphor_amd_traffic_twin -> %d" % rv)
        return ms.value

    def set_overlap(self, enable):
        return self.L.fosphor_amd_set_overlap(self.h, 1 if enable else 0)

    def set_input_ordering(self, strict):
        return self.L.fosphor_amd_set_input_ordering(self.h, 1 if strict else 0)

    def wait_input(self):
        return self.L.fosphor_amd_wait_input(self.h)

    def exchange_time(self):
        """(ms summed over the exchanges recorded while profiling was on, how many); resets"""
        ms, n = C.c_float(), C.c_int()
        rv = self.L.fosphor_amd_exchange_time(self.h, C.byref(ms), C.byref(n))
        if rv:
            raise RuntimeError("fosphor_amd_exchange_time -> %d" % rv)
        return ms.value, n.value

    def share_stats(self):
        """fosphor_amd_share_stats: (FFT launches in the space-sharing form, in the full-chip form, work-groups of the shared form)"""
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_int()
        rv = self.L.fosphor_amd_share_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        if rv:
            raise RuntimeError("fosphor_amd_share_stats -> %d" % rv)
        return a.value, b.value, c.value

    def kernel_busy(self):
        """ms during which >= 1 kernel of each kind ran (call before kernel_times)"""
        ms = (C.c_float * 3)()
        rv = self.L.fosphor_amd_kernel_busy(self.h, C.byref(ms))
        if rv:
            raise RuntimeError("fosphor_amd_kernel_busy -> %d" % rv)
        return list(ms)

    def kernel_times(self):
        ms = (C.c_float * 3)()
        n = (C.c_int * 3)()
        rv = self.L.fosphor_amd_kernel_times(self.h, C.byref(ms), C.byref(n))
        if rv:
            raise RuntimeError("fosphor_amd_kernel_times -> %d" % rv)
        return list(ms), list(n)

    @property
    def stream(self):
        return self.L.fosphor_amd_stream(self.h)

    @property
    def stream2(self):
        return self.L.fosphor_amd_stream2(self.h)
