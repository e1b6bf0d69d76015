"""ctypes binding of gr-fosphor_amd/libfosphor_amd.so (the C ABI of include/fosphor.h and
include/fosphor_amd.h).  No CPU fallback: a missing library is a hard error."""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
# FOSPHOR_AMD_LIB lets the tuning harness (tools/ab_bench.sh) point at an alternative build
LIB_PATH = os.environ.get("FOSPHOR_AMD_LIB") or os.path.join(PKG_DIR, "libfosphor_amd.so")


class Config(C.Structure):
    """struct fosphor_amd_config (include/fosphor_amd.h)"""
    _fields_ = [("fft_len_log", C.c_int), ("n_bins", C.c_int), ("wf_rows", C.c_int),
                ("t0r", C.c_float), ("t0d", C.c_float), ("alpha", C.c_float),
                ("device", C.c_int), ("max_spectra", C.c_int), ("max_batches", C.c_int),
                ("stream", C.c_void_p), ("iq_format", C.c_int)]


class Buffers(C.Structure):
    """struct fosphor_amd_buffers"""
    _fields_ = [("d_waterfall", C.c_void_p), ("d_histogram", C.c_void_p), ("d_spectrum", C.c_void_p),
                ("d_hitcount", C.c_void_p), ("waterfall_pos", C.c_int),
                ("fft_len", C.c_int), ("n_bins", C.c_int), ("wf_rows", C.c_int),
                ("histo_scale", C.c_float), ("histo_offset", C.c_float)]


class Partials(C.Structure):
    """struct fosphor_amd_partials"""
    _fields_ = [("d_hc", C.c_void_p), ("d_live_sum", C.c_void_p), ("d_max", C.c_void_p),
                ("n_hc", C.c_int), ("n_cols", C.c_int)]


class Channel(C.Structure):
    _fields_ = [("enabled", C.c_int), ("center", C.c_float), ("width", C.c_float)]


class Render(C.Structure):
    """struct fosphor_render (include/fosphor.h; layout of the reference's fosphor.h:58-90)"""
    _fields_ = [("pos_x", C.c_int), ("pos_y", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("options", C.c_int), ("histo_wf_ratio", C.c_float), ("freq_n_div", C.c_int),
                ("freq_center", C.c_float), ("freq_span", C.c_float), ("wf_span", C.c_float),
                ("channels", Channel * 8),
                ("_wf_pos", C.c_int), ("_x_div", C.c_float), ("_x", C.c_float * 2), ("_x_label", C.c_float),
                ("_y_histo_div", C.c_float), ("_y_histo", C.c_float * 2), ("_y_wf", C.c_float * 2),
                ("_y_label", C.c_float)]


# every exported entry point: name -> (restype, argtypes)
class FreqAxis(C.Structure):
    """struct fosphor_amd_freq_axis (= the reference's struct freq_axis, axis.h:20-30)"""
    _fields_ = [("center", C.c_double), ("span", C.c_double), ("step", C.c_double), ("mode", C.c_int),
                ("abs_fmt", C.c_char * 16), ("abs_scale", C.c_double), ("rel_fmt", C.c_char * 16),
                ("rel_step", C.c_double)]


SIGNATURES = {
    # include/fosphor.h
    "fosphor_init": (C.c_void_p, []),
    "fosphor_release": (None, [C.c_void_p]),
    "fosphor_process": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_draw": (None, [C.c_void_p, C.POINTER(Render)]),
    "fosphor_set_fft_window_default": (None, [C.c_void_p]),
    "fosphor_set_fft_window": (None, [C.c_void_p, C.c_void_p]),
    "fosphor_set_power_range": (None, [C.c_void_p, C.c_int, C.c_int]),
    "fosphor_set_frequency_range": (None, [C.c_void_p, C.c_double, C.c_double]),
    "fosphor_render_defaults": (None, [C.POINTER(Render)]),
    "fosphor_render_refresh": (None, [C.POINTER(Render)]),
    "fosphor_pos2freq": (C.c_double, [C.c_void_p, C.POINTER(Render), C.c_int]),
    "fosphor_pos2pwr": (C.c_float, [C.c_void_p, C.POINTER(Render), C.c_int]),
    "fosphor_pos2samp": (C.c_int, [C.c_void_p, C.POINTER(Render), C.c_int]),
    "fosphor_freq2pos": (C.c_int, [C.c_void_p, C.POINTER(Render), C.c_double]),
    "fosphor_pwr2pos": (C.c_int, [C.c_void_p, C.POINTER(Render), C.c_float]),
    "fosphor_samp2pos": (C.c_int, [C.c_void_p, C.POINTER(Render), C.c_int]),
    "fosphor_render_pos_inside": (C.c_int, [C.POINTER(Render), C.c_int, C.c_int]),
    # include/fosphor_amd.h
    "fosphor_amd_init": (C.c_void_p, [C.POINTER(Config)]),
    "fosphor_amd_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fosphor_amd_process_device_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fosphor_amd_finish": (C.c_int, [C.c_void_p]),
    "fosphor_amd_get_buffers": (C.c_int, [C.c_void_p, C.POINTER(Buffers)]),
    "fosphor_amd_get_buffers_nohc": (C.c_int, [C.c_void_p, C.POINTER(Buffers)]),
    "fosphor_amd_read": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]),
    "fosphor_amd_fft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_amd_bin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_amd_accumulate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fosphor_amd_accumulate_device_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fosphor_amd_set_partial_slot": (C.c_int, [C.c_void_p, C.c_int]),
    "fosphor_amd_get_partials": (C.c_int, [C.c_void_p, C.POINTER(Partials)]),
    "fosphor_amd_merge": (C.c_int, [C.c_void_p, C.c_int]),
    "fosphor_amd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "fosphor_amd_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "fosphor_amd_comm_destroy": (C.c_int, [C.c_void_p]),
    "fosphor_amd_comm_available": (C.c_int, []),
    "fosphor_amd_comm_count": (C.c_int, [C.c_void_p]),
    "fosphor_amd_exchange_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "fosphor_amd_exchange": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fosphor_amd_exchange_sliced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fosphor_amd_merge_sliced": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fosphor_amd_gather_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fosphor_amd_profile": (None, [C.c_void_p, C.c_int]),
    "fosphor_amd_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 3), C.POINTER(C.c_int * 3)]),
    "fosphor_amd_kernel_busy": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 3)]),
    "fosphor_amd_host_thresholds": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "fosphor_amd_host_twiddle_count": (C.c_int, []),
    "fosphor_amd_host_twiddles": (C.c_int, [C.c_void_p]),
    "fosphor_amd_set_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "fosphor_amd_traffic_twin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "fosphor_amd_set_input_ordering": (C.c_int, [C.c_void_p, C.c_int]),
    "fosphor_amd_wait_input": (C.c_int, [C.c_void_p]),
    "fosphor_amd_stream": (C.c_void_p, [C.c_void_p]),
    "fosphor_amd_stream2": (C.c_void_p, [C.c_void_p]),
    "fosphor_amd_upload_stream": (C.c_void_p, [C.c_void_p]),
    "fosphor_amd_tune_placement": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "fosphor_amd_plan_piece_batches": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong]),
    "fosphor_amd_share_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "fosphor_amd_version": (C.c_char_p, []),
    # include/fosphor_amd_axis.h
    "fosphor_amd_freq_axis_build": (None, [C.c_void_p, C.c_double, C.c_double, C.c_int]),
    "fosphor_amd_freq_axis_render": (None, [C.c_void_p, C.c_char_p, C.c_int]),
    "fosphor_amd_freq_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_amd_power_labels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int * 11)]),
    # include/fosphor_amd_cmap.h
    "fosphor_amd_cmap_generate": (C.c_int, [C.c_int, C.c_void_p, C.c_int]),
    "fosphor_amd_colorize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                       C.c_int, C.c_void_p]),
    # include/fosphor_amd_sink.h
    "fosphor_amd_process_pinned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_amd_upload_pinned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_amd_process_uploaded": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "fosphor_amd_pending_uploads": (C.c_int, [C.c_void_p]),
    "fosphor_amd_wait_upload": (C.c_int, [C.c_void_p]),
    "fosphor_amd_fifo_new": (C.c_void_p, [C.c_int, C.c_int]),
    "fosphor_amd_fifo_free": (None, [C.c_void_p]),
    "fosphor_amd_fifo_free_space": (C.c_int, [C.c_void_p]),
    "fosphor_amd_fifo_used": (C.c_int, [C.c_void_p]),
    "fosphor_amd_fifo_write_max_size": (C.c_int, [C.c_void_p]),
    "fosphor_amd_fifo_write_prepare": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "fosphor_amd_fifo_write_commit": (None, [C.c_void_p, C.c_int]),
    "fosphor_amd_fifo_read_max_size": (C.c_int, [C.c_void_p]),
    "fosphor_amd_fifo_read_peek": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "fosphor_amd_fifo_read_discard": (None, [C.c_void_p, C.c_int]),
    "fosphor_amd_sink_new": (C.c_void_p, []),
    "fosphor_amd_sink_new_len": (C.c_void_p, [C.c_int]),
    "fosphor_amd_sink_feed": (C.c_double, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fosphor_amd_sink_free": (None, [C.c_void_p]),
    "fosphor_amd_sink_dropped": (C.c_uint64, [C.c_void_p]),
    "fosphor_amd_sink_start": (C.c_int, [C.c_void_p]),
    "fosphor_amd_sink_stop": (C.c_int, [C.c_void_p]),
    "fosphor_amd_sink_work": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "fosphor_amd_sink_ui_action": (None, [C.c_void_p, C.c_int]),
    "fosphor_amd_sink_reshape": (None, [C.c_void_p, C.c_int, C.c_int]),
    "fosphor_amd_sink_mouse_action": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "fosphor_amd_sink_set_freq_callback": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fosphor_amd_sink_write_prepare": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "fosphor_amd_sink_write_commit": (None, [C.c_void_p, C.c_int]),
    "fosphor_amd_sink_get_render": (None, [C.c_void_p, C.c_int, C.POINTER(Render)]),
    "fosphor_amd_sink_set_frequency_range": (None, [C.c_void_p, C.c_double, C.c_double]),
    "fosphor_amd_sink_set_fft_window": (None, [C.c_void_p, C.c_void_p]),
    "fosphor_amd_sink_set_visible": (None, [C.c_void_p, C.c_int]),
    "fosphor_amd_sink_core": (C.c_void_p, [C.c_void_p]),
    "fosphor_amd_sink_stats": (None, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fosphor_amd_process_device_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "fosphor_amd_set_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "fosphor_amd_traffic_twin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", ROOT, "all"], stdout=out)
    return LIB_PATH


def load():
    """Load the HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "gr-fosphor_amd: %s is missing -- run `make` (or __graft_entry__.build()); "
                "there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (SONAME
        # libamdhip64.so.7, like /opt/rocm's) but libtorch_hip asks for it by the unversioned
        # file name, so if THIS library is loaded first (pulling in /opt/rocm's copy) a later
        # `import torch` loads a second runtime and device init fails.  Importing torch first
        # makes both bind to the one runtime, so pointers, streams and events are shared.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)		# AttributeError = the library does not export the header's symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
