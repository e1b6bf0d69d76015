"""GPU: gr_fosphor_amd.dist.ShardedFosphor -- two ranks on ONE device over gloo, the native RCCL exchange
(fosphor_amd_exchange: one ncclGroup on the library's stream) on a single rank, and -- where two devices are visible --
between two real devices (test_native_rccl_two_real_devices; skipped on the pool's one-GPU boxes).

The 8-GPU RCCL run is the driver's; this exercises the same rank code (time-sharded accumulate,
per-frame all-reduce of hit counts / live sum / max with the previous frame's exchange left in
flight, merge) with two processes sharing cuda:0 and gloo as transport.  Every rank must end
with hit counts bit-identical to the oracle's for the whole frame, the same histogram /
spectrum on both ranks, and rank 1 (last time block) holding the surviving waterfall rows."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["FOSPHOR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FOSPHOR_ROOT"], "tests"))
import torch, torch.distributed as dist
from _pkg import gr_fosphor_amd
from gr_fosphor_amd.dist import ShardedFosphor, shard_range
from oracle_lib import Oracle, gaussian_iq, add_tone, digest

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
N = 1024
frames = [2048, 2048, 1024]		# spectra per frame (all ranks together)
sf = ShardedFosphor(gr_fosphor_amd.Fosphor, rank, world, exchange="torch", max_spectra=2048)	# gloo carries the arrays
o = Oracle()
t0 = 0
for k, total in enumerate(frames):
    x = add_tone(gaussian_iq(total * N, 70 + k), 0.1, 0.11 + 0.02 * k, t0=t0)
    t0 += total * N
    off, n = shard_range(total, rank, world)
    d = torch.from_numpy(x[off * N:(off + n) * N]).cuda()
    sf.frame(d, total, overlap=True)			# exchange of frame k left in flight
    assert o.process(x, strict=False, nthreads=4) == 0
sf.flush()
f = sf.f
assert f.finish() >= 0
assert f.waterfall_pos == o.waterfall_pos
assert np.array_equal(f.hitcount, o.hitcount.T), "rank %d: hit counts differ from the oracle" % rank
h, s = f.histogram, f.spectrum
assert np.allclose(h, o.histogram, rtol=1e-4, atol=2e-6), "rank %d histogram" % rank
assert np.allclose(s[..., 1], o.spectrum[..., 1], rtol=1e-4, atol=1e-6), "rank %d spectrum" % rank
# replicated state: both ranks hold the same bits
mine = torch.tensor([int(digest(h)[:15], 16), int(digest(s)[:15], 16)], dtype=torch.int64)
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(b, both[0]) for b in both), "persistent state differs between ranks"
if rank == world - 1:		# the last 1024 spectra of the last frame belong to ... both halves: check own rows
    off, n = shard_range(frames[-1], rank, world)
    rows = (o.waterfall_pos - frames[-1] + off + np.arange(n)) & 1023
    assert np.allclose(f.waterfall[rows], o.waterfall[rows], rtol=1e-4, atol=1e-6)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def test_two_rank_frames_on_one_gpu(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FOSPHOR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
        assert "rank %d ok" % r in out


NATIVE = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["FOSPHOR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FOSPHOR_ROOT"], "tests"))
import torch
from _pkg import gr_fosphor_amd
from gr_fosphor_amd.dist import ShardedFosphor
from oracle_lib import Oracle, gaussian_iq, add_tone

# One rank, the library's OWN RCCL communicator (no torch.distributed at all): accumulate -> ncclGroup of three
# all-reduces on the count/merge stream -> merge, several frames back to back without host synchronisation.
torch.cuda.set_device(0)
N = 1024
sliced = os.environ.get("FOSPHOR_TEST_SLICED") == "1"
frames = [2048, 1024, 4096, 2048]
sf = ShardedFosphor(gr_fosphor_amd.Fosphor, 0, 1, exchange="rccl", force_exchange=True, sliced=sliced, max_spectra=4096)
assert sf.comm is not None and sf.sliced == sliced
o = Oracle()
t0 = 0
keep = []
for k, total in enumerate(frames):
    x = add_tone(gaussian_iq(total * N, 170 + k), 0.1, 0.09 + 0.02 * k, t0=t0)
    t0 += total * N
    keep.append(torch.from_numpy(x).cuda())
    sf.frame(keep[-1], total)
    assert o.process(x, strict=False, nthreads=4) == 0
sf.gather_state()
f = sf.f
assert f.finish() >= 0
assert f.waterfall_pos == o.waterfall_pos
assert np.array_equal(f.hitcount, o.hitcount.T), "hit counts differ from the oracle"
assert np.allclose(f.histogram, o.histogram, rtol=1e-4, atol=2e-6)
assert np.allclose(f.spectrum[..., 1], o.spectrum[..., 1], rtol=1e-4, atol=1e-6)
rows = (o.waterfall_pos - 1024 + np.arange(1024)) & 1023
assert np.allclose(f.waterfall[rows], o.waterfall[rows], rtol=1e-4, atol=1e-6)
sf.close()
print("native ok")
'''


@pytest.mark.parametrize("sliced", ["0", "1"])
def test_native_rccl_exchange_single_rank(tmp_path, sliced):
    """fosphor_amd_exchange / fosphor_amd_exchange_sliced on a real RCCL communicator (world size 1: the
    8-GPU run is the driver's): stream ordering K2 -> ncclGroup -> K3 without host waits, library-owned
    buffers, reduce-scatter + sliced merge + all-gather; state equal to the oracle's after four frames."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    script = tmp_path / "native.py"
    script.write_text(NATIVE)
    env = dict(os.environ, FOSPHOR_ROOT=ROOT, FOSPHOR_TEST_SLICED=sliced)
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert p.returncode == 0 and "native ok" in p.stdout, p.stdout[-3000:]


NATIVE2 = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["FOSPHOR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FOSPHOR_ROOT"], "tests"))
import torch, torch.distributed as dist
from _pkg import gr_fosphor_amd
from gr_fosphor_amd.dist import ShardedFosphor, shard_range
from oracle_lib import Oracle, gaussian_iq, add_tone, digest

# TWO REAL DEVICES, one process each, the library's own RCCL communicator over xGMI: gloo only carries rank 0's 128-byte id and the
# digests compared at the end.
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)
N = 1024
sliced = os.environ.get("FOSPHOR_TEST_SLICED") == "1"
frames = [2048, 1024, 4096, 2048]
sf = ShardedFosphor(gr_fosphor_amd.Fosphor, rank, world, exchange="rccl", sliced=sliced, max_spectra=4096)
assert sf.comm is not None and sf.sliced == sliced
assert sf.exchange_ranks() == world, "ncclCommCount says %d ranks" % sf.exchange_ranks()
o = Oracle()
t0 = 0
keep = []
for k, total in enumerate(frames):
    x = add_tone(gaussian_iq(total * N, 270 + k), 0.1, 0.07 + 0.02 * k, t0=t0)
    t0 += total * N
    off, n = shard_range(total, rank, world)
    keep.append(torch.from_numpy(x[off * N:(off + n) * N]).cuda())
    sf.frame(keep[-1], total)				# accumulate -> exchange -> merge: three asynchronous C calls
    assert o.process(x, strict=False, nthreads=4) == 0
sf.gather_state()
f = sf.f
assert f.finish() >= 0
assert f.waterfall_pos == o.waterfall_pos
assert np.array_equal(f.hitcount, o.hitcount.T), "rank %d: hit counts differ from the oracle" % rank
h, s = f.histogram, f.spectrum
assert np.allclose(h, o.histogram, rtol=1e-4, atol=2e-6), "rank %d histogram" % rank
assert np.allclose(s[..., 1], o.spectrum[..., 1], rtol=1e-4, atol=1e-6), "rank %d spectrum" % rank
mine = torch.tensor([int(digest(h)[:15], 16), int(digest(s)[:15], 16), int(digest(f.hitcount)[:15], 16)], dtype=torch.int64)
both = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(b, both[0]) for b in both), "replicated state differs between the ranks"
off, n = shard_range(frames[-1], rank, world)		# the waterfall rows this rank computed
rows = (o.waterfall_pos - frames[-1] + off + np.arange(n)) & 1023
assert np.allclose(f.waterfall[rows], o.waterfall[rows], rtol=1e-4, atol=1e-6)
dist.barrier()
sf.close()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


@pytest.mark.parametrize("sliced", ["0", "1"])
def test_native_rccl_two_real_devices(tmp_path, sliced):
    """The native RCCL exchange between TWO REAL DEVICES (north_star: per-display-frame RCCL all-reduce over xGMI), both
    forms -- one ncclGroup of three all-reduces, and reduce-scatter + frequency-sliced merge + all-gather -- with the assertions of
    the gloo test on the real transport: hit counts array_equal to the oracle's on every rank, histogram / spectrum in tolerance,
    the replicated state bit-identical across the ranks, ncclCommCount == 2.  Skipped (reason printed) where fewer than two
    devices are visible: the pool's boxes have one, so the first machine with two runs it."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices for a real 2-rank RCCL exchange; this box shows %d" % torch.cuda.device_count())
    script = tmp_path / "native2.py"
    script.write_text(NATIVE2)
    env = dict(os.environ, FOSPHOR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", WORLD_SIZE="2", FOSPHOR_TEST_SLICED=sliced)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
        assert "rank %d ok" % r in out


def test_c_program_native_exchange(tmp_path):
    """tests/c/exchange_test.c: a plain C host (no Python, no PyTorch in the process) links libfosphor_amd.so, creates the
    library's RCCL communicator and runs accumulate -> fosphor_amd_exchange -> merge for three frames; counts and histogram
    must equal the single-launch path of the same ABI."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    exe = tmp_path / "exchange_test"
    libdir = os.path.join(ROOT, "gr-fosphor_amd")
    cmd = ["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "exchange_test.c"),
           "-o", str(exe), "-L", libdir, "-lfosphor_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert b.returncode == 0, b.stdout
    p = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "c exchange ok" in p.stdout, p.stdout[-2000:]


def test_bench_frame_mode_two_ranks_one_gpu(tmp_path):
    """bench.py's N > 1 path end to end -- process-group set-up, transport agreement, time-sharded frames of 256 batches per rank,
    exchange once per frame, barrier + max-over-ranks timing, ONE JSON line from rank 0 -- with two ranks sharing this GPU.
    RCCL refuses two ranks on one device, so the ranks talk over gloo (FOSPHOR_BENCH_BACKEND, a test hook) and the exchange is the
    torch transport; what is checked is the bench's own logic, not a rate."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29647", WORLD_SIZE="2", LOCAL_RANK="0",
               FOSPHOR_BENCH_BACKEND="gloo", FOSPHOR_AMD_EXCHANGE="torch")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--ring-steps", "1",
           "--precondition", "0.05", "--no-cpu-baseline", "--no-extra-passes", "--no-traffic-twin"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        outs.append((out, err))
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s\n%s" % (r, out[-2000:], err[-3000:])
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")], "exactly one JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak"
    assert j["config"]["mode"] == "frame" and "torch.distributed" in j["config"]["exchange"]
    chk = j["config"]["exchange_check"]		# the line's own check of what the exchange left on the ranks
    assert chk["every_spectrum_counted_once_on_every_rank"] and chk["replicated_state_bit_identical_across_ranks"], chk
    assert chk["hit_counts_per_column"] == 2 * 256 * 1024
    # whole-job aggregate: both ranks' samples over the slowest rank's time
    assert abs(j["value"] * 1e6 * j["ms_per_step"] * 1e-3 - 2 * 256 * 1024 * 1024) < 1e3
    assert j["value"] > 0 and j["roofline"]["frac"] > 0


@pytest.mark.parametrize("world,extra", [(2, []), (8, ["--batches-per-step", "16"])])
def test_bench_self_launch_ranks_on_one_gpu(world, extra):
    """`python3 bench.py --gpus N` with NO launcher in the environment: bench.py itself starts torch.distributed.run as a child
    process (before any GPU call in the parent), relays rank 0's one JSON line and the child's exit code.  The N ranks share this
    GPU (test hooks: FOSPHOR_BENCH_ONE_GPU puts every rank on device 0, the ranks talk over gloo as in the test above).  N = 8 is
    the driver's largest run: launcher, process group, transport agreement, the 8-way time split of a frame, the max-over-ranks
    timing and the one JSON line are exercised at that world size before the first real 8-GPU contact."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FOSPHOR_BENCH_BACKEND="gloo", FOSPHOR_AMD_EXCHANGE="torch", FOSPHOR_BENCH_ONE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--ring-steps", "1",
           "--precondition", "0.05", "--no-cpu-baseline", "--no-extra-passes", "--no-traffic-twin"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, "%s\n%s" % (p.stdout[-2000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 3 and j["config"]["mode"] == "frame" and j["value"] > 0
    assert len(j["config"]["k1_busy_ms_per_launch_per_rank"]["all"]) == world
    assert "torch.distributed.run" in p.stderr
