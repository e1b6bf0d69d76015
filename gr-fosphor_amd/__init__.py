"""gr-fosphor_amd: MI355X-native compute core of the fosphor spectrum display.

The directory name carries a hyphen (it mirrors the reference's repository name), so import
it through the loader at the repository root:

    from _pkg import gr_fosphor_amd
    f = gr_fosphor_amd.Fosphor()

Contents: csrc/ (HIP kernels + the C ABI of include/*.h), _lib.py (ctypes binding),
core.py (host-side mirror of the reference's libfosphor interface), dist.py (multi-GPU frame).
"""
from . import _lib
from ._lib import LIB_PATH, Buffers, Config, Partials, Render, build, load
from .core import FFT_LEN, FFT_LEN_LOG, MAX_BATCH, MULT_BATCH, Fosphor

__all__ = ["Fosphor", "build", "load", "LIB_PATH", "Config", "Buffers", "Partials", "Render",
           "FFT_LEN", "FFT_LEN_LOG", "MAX_BATCH", "MULT_BATCH"]
