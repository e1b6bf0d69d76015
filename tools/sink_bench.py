#!/usr/bin/env python3
"""PCIe-inclusive streaming rate through the GNU-Radio-free sink: a native thread calls sink_runtime::work()
(fosphor_amd_sink_feed), the worker thread DMAs out of the pinned FIFO with several regions in flight.

    python3 tools/sink_bench.py [fifo_log2 ...]        (default: 21 = the reference's 2 Mi samples, and 24)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from _pkg import gr_fosphor_amd  # noqa: E402

L = gr_fosphor_amd.load()
n = 32 << 20						# 32 Mi samples = 256 MiB per pass
x = (np.random.default_rng(3).standard_normal((n, 2)) * 0.05).astype(np.float32)
for lg in [int(a) for a in sys.argv[1:]] or [21, 24]:
    for chunk in (64 * 1024, 1 << 20):
        s = L.fosphor_amd_sink_new_len(1 << lg)
        assert L.fosphor_amd_sink_start(s) == 1
        L.fosphor_amd_sink_feed(s, x.ctypes.data, n, chunk, 1)		# warm-up (boot, staging buffers)
        reps = 8
        dt = L.fosphor_amd_sink_feed(s, x.ctypes.data, n, chunk, reps)
        frames = C.c_uint64()
        L.fosphor_amd_sink_stats(s, C.byref(frames), None, None, None, None)
        print("fifo 2^%d samples, work() calls of %7d samples: %6.2f GSamples/s (%.1f GB/s), %d frames drawn"
              % (lg, chunk, n * reps / dt / 1e9, n * reps * 8 / dt / 1e9, frames.value))
        L.fosphor_amd_sink_stop(s)
        L.fosphor_amd_sink_free(s)
