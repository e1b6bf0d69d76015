"""CPU: the drop-in boundary without a GPU -- the C-ABI library loads, exports every symbol the
headers declare, refuses to run without a device (no CPU fallback), and its host-side tables
(bin thresholds, twiddles) agree with the oracle.  No compute kernels are launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle_lib import Oracle, oracle_bins

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd():
    from _pkg import gr_fosphor_amd
    if not os.path.exists(gr_fosphor_amd.LIB_PATH):
        gr_fosphor_amd.build()		# hipcc cross-compiles gfx950 without a GPU
    gr_fosphor_amd.load()
    return gr_fosphor_amd


def declared_functions(header):
    """function names declared in a header (crude but sufficient for these plain-C headers)"""
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(fosphor_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(amd):
    lib = C.CDLL(amd.LIB_PATH)
    want = sorted(set(declared_functions("fosphor.h") + declared_functions("fosphor_amd.h")
                      + declared_functions("fosphor_amd_sink.h") + declared_functions("fosphor_amd_cmap.h")
                      + declared_functions("fosphor_amd_axis.h")))
    want = [n for n in want if n not in ("fosphor_amd_fifo", "fosphor_amd_sink", "fosphor_render")]	# type names
    assert "fosphor_process" in want and "fosphor_amd_process_device" in want and len(want) >= 30
    missing = [n for n in want if not hasattr(lib, n)]
    assert not missing, "declared but not exported: %s" % missing
    # and the ctypes binding covers them all
    from gr_fosphor_amd import _lib
    unbound = [n for n in want if n not in _lib.SIGNATURES]
    assert not unbound, "exported but unbound in _lib.py: %s" % unbound


def test_product_library_has_no_wrong_result_switches(amd):
    """The measurement switches that produce wrong spectra by construction (FOSPHOR_AMD_DBG_*: skip a kernel, alias chunks, drop
    a wait, CU masks) exist only in probe builds (-DFOSPHOR_AMD_PROBES, tools/r04_ceiling_build.sh).  The shipped library must not
    contain the name of any of them: a stray environment variable cannot change its results."""
    blob = open(amd.LIB_PATH, "rb").read()
    assert b"FOSPHOR_AMD_DBG" not in blob
    assert b"FOSPHOR_AMD_TILE" in blob			# (the check can see environment names: a tuning knob that IS read, once, at init)


def test_piece_planner_host_logic(amd):
    """fosphor_amd_plan_piece_batches: how a device-resident call is cut into sub-launches (pure host arithmetic).  Pieces are
    about sub_samples samples and equal; at fft_len_log = 13 with the streams on they are whole multiples of the unit that makes
    a piece's tiles (64 spectra each) a multiple of 224, so that the FFT launch can leave CUs to the count / merge kernels
    (DESIGN.md sections 4-5; DESIGN_HISTORY.md section 8) -- when the call's batch count allows it."""
    plan = amd.load().fosphor_amd_plan_piece_batches
    assert plan(10, 1, 256, 1024, 1 << 26) == 64		# the C2 bench call: 4 pieces of 64 reference batches
    assert plan(10, 1, 100, 1024, 1 << 26) == 50		# equal pieces, not 64 + 36
    assert plan(10, 1, 3, 1024, 1 << 26) == 3
    assert plan(13, 1, 28, 4096, 1 << 30) == 28			# the C3 bench call: one piece (1792 tiles = 8 x 224)
    assert plan(13, 1, 56, 4096, 1 << 30) == 28			# cap 32 -> whole units of 7 batches
    assert plan(13, 1, 64, 4096, 1 << 30) == 32			# 64 is no multiple of 7: plain equal pieces
    assert plan(13, 0, 56, 4096, 1 << 30) == 28			# one stream: equal pieces (28 + 28), no unit logic needed
    assert plan(13, 1, 28, 1024, 1 << 27) == 14			# 1024-spectrum batches are 16 tiles: unit 14, cap 16
    assert plan(13, 1, 42, 1024, 1 << 27) == 14
    assert plan(16, 1, 8, 1024, 1 << 26) == 1			# a 65536-point frame is a piece of its own
    assert plan(13, 1, 4, 4096, 1 << 20) == 1			# a piece is never less than a batch
    for bad in [(0, 1, 4, 64, 1 << 26), (10, 1, 0, 64, 1 << 26), (10, 1, 4, 0, 1 << 26), (10, 1, 4, 64, 0)]:
        assert plan(*bad) == -22				# -EINVAL


def test_struct_layouts_match_reference_abi(amd):
    """struct fosphor_render / fosphor_channel layout (fosphor.h:42-90 of the reference):
    10 user words, 8 channels of 3 words, then 11 private words."""
    assert C.sizeof(amd.Render) == 4 * (10 + 8 * 3 + 11)
    assert amd.Render.channels.offset == 40
    assert amd.Render._wf_pos.offset == 40 + 96
    assert amd.Render._y_label.offset == C.sizeof(amd.Render) - 4


def test_no_device_no_fallback(amd, capfd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = amd.load()
    assert L.fosphor_init() is None			# NULL, like the reference on failure (fosphor.c:70-73)
    err = capfd.readouterr().err
    assert "no CPU path" in err or "no HIP device" in err
    with pytest.raises(RuntimeError):
        amd.Fosphor()


def test_render_geometry_helpers(amd):
    """fosphor_render_defaults / refresh / pos mapping are pure CPU maths (fosphor.c:162-387)."""
    L = amd.load()
    r = amd.Render()
    L.fosphor_render_defaults(C.byref(r))
    assert (r.width, r.height, r.freq_n_div) == (1024, 1024, 10)
    assert r.options == 0x17f and abs(r.histo_wf_ratio - 0.5) < 1e-9
    L.fosphor_render_refresh(C.byref(r))
    # width 1024: reserved 10+30 left, 10+10 right -> 964 usable -> 10 divisions of 96 px, 4 px slack
    assert r.freq_n_div == 10 and r._x_div == 96.0
    assert r._x[0] == 40.0 + 2.0 and r._x[1] == r._x[0] + 960.0 + 1.0
    # height 1024 with waterfall + freq labels: (1024 - 40) * 0.5 = 492 -> 49 px divisions
    assert r._y_histo_div == 49.0
    assert r._y_histo[1] == 1014.0 and r._y_histo[0] == 1014.0 - 490.0 - 1.0
    assert r._y_wf[0] == 10.0 and r._y_wf[1] == r._y_histo[0] - 10.0 - 10.0
    inside = L.fosphor_render_pos_inside(C.byref(r), 500, 800)
    assert inside == 3
    assert L.fosphor_render_pos_inside(C.byref(r), 500, 100) == 5
    assert L.fosphor_render_pos_inside(C.byref(r), 5, 5) == 0


def test_bin_thresholds_reproduce_oracle_bins(amd, oracle_built):
    """The table the GPU compares against must reproduce the oracle's
    log10(hypot()) -> round() pipeline for any sample: count(s >= thr[b]) == oracle bin."""
    L = amd.load()
    rng = np.random.default_rng(3)
    for n_bins, (db_ref, db_div) in [(128, (0, 10)), (256, (0, 10)), (128, (-20, 5)), (256, (10, 2))]:
        o = Oracle(n_bins=n_bins)
        o.set_power_range(db_ref, db_div)
        thr = np.empty(n_bins + 1, np.float64)
        assert L.fosphor_amd_host_thresholds(n_bins, o.histo_scale, o.histo_offset, thr.ctypes.data) == 0
        assert thr[0] == -1.0 and np.all(np.diff(thr[1:]) >= 0)
        # random samples + samples straddling every threshold by one ulp of the double
        mag = np.exp(rng.uniform(np.log(1e-12), np.log(1e12), 200000))
        ph = rng.uniform(0, 2 * np.pi, mag.size)
        v = np.stack([mag * np.cos(ph), mag * np.sin(ph)], 1).astype(np.float32)
        want, _ = oracle_bins(v, o.histo_scale, o.histo_offset, n_bins)
        s = v[:, 0].astype(np.float64) ** 2 + v[:, 1].astype(np.float64) ** 2
        got = np.searchsorted(thr[1:n_bins], s, side="right")
        got = np.where(s >= thr[n_bins], 0, got)
        assert np.array_equal(got, want)
        # exactly at / just below each threshold, with (re, im) = (sqrt-free) axis samples:
        # h = float; s = h*h exactly in double
        for b in range(1, n_bins):
            if thr[b] >= thr[n_bins]:
                continue
            h = np.float32(np.sqrt(thr[b]))
            cand = np.array([np.nextafter(h, np.float32(0)), h, np.nextafter(h, np.float32(np.inf))], np.float32)
            vv = np.stack([cand, np.zeros(3, np.float32)], 1)
            w, _ = oracle_bins(vv, o.histo_scale, o.histo_offset, n_bins)
            ss = cand.astype(np.float64) ** 2
            g = np.searchsorted(thr[1:n_bins], ss, side="right")
            assert np.array_equal(g, w), (n_bins, b)


def test_twiddle_table_matches_oracle(amd, oracle_built):
    L = amd.load()
    OL = Oracle.lib()
    n = L.fosphor_amd_host_twiddle_count()
    assert n == 8 * 7 + 64 * 7 + 512
    tw = np.empty((n, 2), np.float32)
    assert L.fosphor_amd_host_twiddles(tw.ctypes.data) == 0
    cs = np.empty(2, np.float32)
    i = 0
    for p in (8, 64):
        for k in range(p):
            for f in range(1, 8):
                OL.fosphor_oracle_twiddle(0, p, k, f, cs.ctypes.data)
                assert np.array_equal(tw[i].view(np.uint32), cs.view(np.uint32)), (p, k, f)
                i += 1
    for k in range(512):
        OL.fosphor_oracle_twiddle(1, 512, k, 1, cs.ctypes.data)
        assert np.array_equal(tw[i].view(np.uint32), cs.view(np.uint32)), k
        i += 1
    assert i == n


def test_k1w_hand_issued_requests_are_not_touched_before_their_wait():
    """The 8192-point kernel requests the next spectrum's IQ with inline-assembly buffer loads and waits for them with a hand-written
    s_waitcnt one iteration later (the compiler's own wait also sat through every index store issued in between).  The compiler does not
    know those registers are in flight: the compiled code must not read or write them between a request and the wait, in any instantiation
    (tools/check_k1w_loads.py, on `hipcc -S` of the kernel file)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_k1w_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("untouched until the wait") == 5, r.stdout		# overlap 2, 4, 8, 16 and the general form
    # ... the counted wait's immediate equals the index stores an odd spectrum's epilogue issues behind the requests, and nothing spills
    # inside the spectrum loop's product path (the kernel sits at 256 registers; ScratchSize is printed, budget 64 bytes)
    assert r.stdout.count("vmcnt(16) = 16 index stores per wave") == 5 and r.stdout.count("no spill in the loop's product path") == 5, r.stdout


def test_product_never_touches_the_oracle():
    """The product tree must not import, link or open anything under oracle/ (only tests,
    smoke() and bench's cpu_baseline may)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "gr-fosphor_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(base, fn)).read()
                if re.search(r'#include\s*[<"][^>"]*oracle|import\s+oracle_lib|from\s+oracle_lib|'
                             r'libfosphor_oracle|libfosphor_ref|fosphor_oracle_[a-z]+\s*\(', txt):
                    bad.append(os.path.join(base, fn))
    assert not bad, bad


def test_fifo_semantics(amd):
    """lib/fifo.{h,cc}: power-of-two ring, one slot kept empty, contiguous regions to the end of
    the ring, non-blocking prepare / peek return NULL."""
    import threading
    L = amd.load()
    f = L.fosphor_amd_fifo_new(1024, 0)
    assert f and L.fosphor_amd_fifo_new(1000, 0) is None		# power of two only
    assert L.fosphor_amd_fifo_used(f) == 0 and L.fosphor_amd_fifo_free_space(f) == 1023	# fifo.cc:28-38
    assert L.fosphor_amd_fifo_write_max_size(f) == 1024 and L.fosphor_amd_fifo_read_max_size(f) == 1024
    assert L.fosphor_amd_fifo_read_peek(f, 1, 0) is None			# empty, no wait
    p = L.fosphor_amd_fifo_write_prepare(f, 600, 0)
    buf = (C.c_float * 1200).from_address(p)
    buf[0], buf[1199] = 1.5, -2.5
    L.fosphor_amd_fifo_write_commit(f, 600)
    assert L.fosphor_amd_fifo_used(f) == 600 and L.fosphor_amd_fifo_free_space(f) == 423
    assert L.fosphor_amd_fifo_write_max_size(f) == 424			# distance to the end of the ring
    assert L.fosphor_amd_fifo_write_prepare(f, 424, 0) is None		# only 423 free
    q = L.fosphor_amd_fifo_read_peek(f, 600, 0)
    rb = (C.c_float * 1200).from_address(q)
    assert rb[0] == 1.5 and rb[1199] == -2.5
    L.fosphor_amd_fifo_read_discard(f, 600)
    assert L.fosphor_amd_fifo_used(f) == 0 and L.fosphor_amd_fifo_read_max_size(f) == 424
    # wrap: fill to the end, then the write pointer is back at 0
    L.fosphor_amd_fifo_write_prepare(f, 424, 0); L.fosphor_amd_fifo_write_commit(f, 424)
    assert L.fosphor_amd_fifo_write_max_size(f) == 1024 and L.fosphor_amd_fifo_used(f) == 424
    # blocking write is released by a reader
    L.fosphor_amd_fifo_write_prepare(f, 500, 0); L.fosphor_amd_fifo_write_commit(f, 500)	# used 924
    done = []
    def writer():
        L.fosphor_amd_fifo_write_prepare(f, 200, 1)				# needs 200 free, has 99: blocks
        done.append(1)
    t = threading.Thread(target=writer); t.start()
    t.join(0.2); assert t.is_alive() and not done
    L.fosphor_amd_fifo_read_discard(f, 424)
    t.join(2.0); assert done
    L.fosphor_amd_fifo_free(f)


def test_fifo_two_threads_under_thread_sanitizer(amd, tmp_path):
    """The sink's FIFO is a single-producer / single-consumer ring on two atomic counters (lock-free unless a side has to
    sleep).  tests/c/fifo_stress.cpp runs a producer and a consumer with random region sizes, blocking, non-blocking and timed
    calls, checks that every sample arrives once and in order, and is built with -fsanitize=thread (CPU only: the sanitizer
    instruments fosphor_sink.cpp itself; the rest of the library is linked as it is)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = tmp_path / "fifo_stress"
    libdir = os.path.dirname(amd.LIB_PATH)
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "c", "fifo_stress.cpp"), os.path.join(ROOT, "gr-fosphor_amd", "csrc", "fosphor_sink.cpp"),
           "-L" + libdir, "-lfosphor_amd", "-L/opt/rocm/lib", "-lamdhip64", "-pthread",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "tsan" in (b.stderr or "").lower():
        pytest.skip("ThreadSanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe), "3000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert "0 bad" in r.stdout


def test_bench_self_launch_is_a_child_process_and_relays_failure():
    """`python3 bench.py --gpus N` with no launcher (WORLD_SIZE unset) starts `python -m torch.distributed.run --nproc-per-node N
    bench.py ...` as a CHILD before anything in the parent touches the GPU (bench.py imports torch only behind that branch) and
    exits with the child's code.  Here, without a GPU, every rank fails at torch.cuda.set_device: the parent must exit non-zero
    without a JSON line -- never zero, never a hang."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "3"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[4:6] == ["--nproc-per-node", "8"]
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node 2" in p.stderr, p.stderr[-2000:]
    assert p.returncode != 0, "no GPU here: the launched ranks cannot have succeeded"
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_scaling_record_without_the_native_exchange():
    """bench.py on N > 1 ranks: a line whose exchange was not the native RCCL transport spanning all N ranks (ncclCommCount on every
    rank) is not a scaling record -- the run is marked `invalid` and exits 3 instead of reporting a rate the driver would divide by
    N (no silent fallback).  The guard as a pure function; the gloo-on-one-GPU tests pass the hook that says the fallback is meant."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    native = "native RCCL (library communicator)"
    g = bench.scaling_record_guard
    assert g(1, "none", [1]) is None						# one rank: nothing to guard
    assert g(8, native, [8] * 8) is None
    assert g(2, native, [2, 2]) is None
    assert "not the native RCCL transport" in g(8, "torch.distributed (nccl)", [8] * 8)
    assert "not the native RCCL transport" in g(2, "none", [1, 1])
    assert "not 8 on every rank" in g(8, native, [8, 8, 8, 1, 8, 8, 8, 8])		# one rank's communicator is short
    assert "not 8 on every rank" in g(8, native, [8] * 7)				# a rank did not report
    assert "not 2 on every rank" in g(2, native, [1, 1])
    assert g(2, "torch.distributed (gloo)", [2, 2], test_hook=True) is None		# tests/test_gpu_dist.py: meant
    # ... and the line's own check of what the exchange left (every spectrum counted once per column on every rank, the replicated state
    # bit-identical across ranks): wrong results fail the run whatever the transport
    good = {"every_spectrum_counted_once_on_every_rank": True, "replicated_state_bit_identical_across_ranks": True}
    assert g(8, native, [8] * 8, check=good) is None
    assert "wrong results" in g(8, native, [8] * 8, check=dict(good, replicated_state_bit_identical_across_ranks=False))
    assert "wrong results" in g(2, "torch.distributed (gloo)", [2, 2], test_hook=True, check=dict(good, every_spectrum_counted_once_on_every_rank=False))
