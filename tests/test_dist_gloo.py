"""CPU, world_size 2, gloo: the per-frame exchange of the multi-GPU path.

Each rank holds the partial arrays of its time-shard of one batch (here produced by the CPU
oracle, since there is no GPU): integer hit counts, weighted live sum, max.  After
gr_fosphor_amd.dist.allreduce_partials every rank must hold exactly the counts of the whole
batch (bit-exact: integer sums are order-independent), the same max, and the live sum within
float tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["FOSPHOR_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FOSPHOR_ROOT"], "tests"))
import torch, torch.distributed as dist
from _pkg import gr_fosphor_amd
from gr_fosphor_amd.dist import allreduce_partials, shard_range, combine_partials_numpy
from oracle_lib import Oracle, gaussian_iq, add_tone

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B, N, alpha = 64, 1024, 0.002
x = add_tone(gaussian_iq(B * N, 99), 0.2, 0.17)

def partials(samples, t_off, total):
    o = Oracle()
    assert o.process(samples) == 0
    n = samples.shape[0] // N
    rows = o.waterfall[:n].astype(np.float64)
    w = (1.0 - np.float32(alpha)).astype(np.float64) ** (total - 1 - (t_off + np.arange(n)))
    return o.hitcount.T.copy(), (rows * w[:, None]).sum(0).astype(np.float32), rows.max(0).astype(np.float32)

off, n = shard_range(B, rank, world)
hc, live, vmax = partials(x[off * N:(off + n) * N], off, B)
t_hc = torch.from_numpy(hc.astype(np.int32).reshape(-1))
t_live = torch.from_numpy(live.copy()); t_max = torch.from_numpy(vmax.copy())
allreduce_partials(t_hc, t_live, t_max)

hc_f, live_f, max_f = partials(x, 0, B)
assert np.array_equal(t_hc.numpy().reshape(hc_f.shape).astype(np.uint32), hc_f), "counts not bit-exact"
assert int(t_hc.sum()) == B * N
assert np.array_equal(t_max.numpy(), max_f), "max differs"
assert np.allclose(t_live.numpy(), live_f, rtol=1e-5, atol=1e-6), "live sum differs"

# the host combination rule agrees too
parts = [partials(x[o_ * N:(o_ + n_) * N], o_, B) for o_, n_ in (shard_range(B, r, world) for r in range(world))]
c_hc, c_live, c_max = combine_partials_numpy(parts)
assert np.array_equal(c_hc, hc_f) and np.array_equal(c_max, max_f)

# async form returns handles and leaves the same result
t2 = torch.from_numpy(hc.astype(np.int32).reshape(-1)); l2 = torch.from_numpy(live.copy()); m2 = torch.from_numpy(vmax.copy())
for wk in allreduce_partials(t2, l2, m2, async_op=True):
    wk.wait()
assert torch.equal(t2, t_hc) and torch.equal(m2, t_max)
try:
    shard_range(48, 0, 2 if world == 2 else world)   # 48 / 2 = 24: not a whole number of 16-spectrum groups
    raise SystemExit("shard_range accepted a ragged split")
except ValueError:
    pass
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def test_allreduce_partials_world2_gloo(oracle_built, tmp_path):
    from _pkg import gr_fosphor_amd
    if not os.path.exists(gr_fosphor_amd.LIB_PATH):
        gr_fosphor_amd.build()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FOSPHOR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617",
               WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
        assert "rank %d ok" % r in out


TRANSPORT_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FOSPHOR_ROOT"])
import torch, torch.distributed as dist
from _pkg import gr_fosphor_amd
from gr_fosphor_amd.dist import agree_on_transport, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
fail_on = int(os.environ["FAIL_ON"])          # rank whose native set-up fails (-1: none)

class Obj:
    def __init__(self, kind): self.kind, self.closed = kind, False
    def close(self): self.closed = True

made = []
def native():
    if rank == fail_on:
        raise RuntimeError("fosphor_amd_comm_init -> -38")     # what NativeComm raises without an RCCL library
    o = Obj("native"); made.append(o); return o
def fallback():
    o = Obj("torch"); made.append(o); return o
def all_reduce_min(v):
    t = torch.tensor([v], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())

logs = []
obj, which = agree_on_transport(native, fallback, world, all_reduce_min, log=logs.append)
want = "native" if fail_on < 0 else "fallback"
assert which == want, (rank, which)
assert obj.kind == ("native" if fail_on < 0 else "torch")
if fail_on >= 0 and rank != fail_on:
    # this rank's native object was built, then closed again when the ranks agreed to fall back together
    assert [o.kind for o in made] == ["native", "torch"] and made[0].closed and not made[1].closed
if rank == fail_on:
    assert logs and "unavailable" in logs[0]
# NativeComm's id hand-off: rank 0's 128-byte id reaches every rank; if rank 0 cannot make one, EVERY rank raises (nobody is
# left waiting in the broadcast)
from gr_fosphor_amd.dist import NativeComm
class FakeLib:
    def __init__(self, fail): self.fail, self.seen = fail, None
    def fosphor_amd_comm_unique_id(self, buf):
        if self.fail: return -38
        buf.raw = bytes(range(128)); return 0
    def fosphor_amd_comm_init(self, href, world, rank, ident):
        self.seen = bytes(ident); return 0
    def fosphor_amd_comm_destroy(self, h): return 0
fl = FakeLib(False)
nc = NativeComm(fl, rank, world)
assert fl.seen == bytes(range(128)), "rank %d got a different id" % rank
nc.h = None
try:
    NativeComm(FakeLib(True), rank, world)
    raise SystemExit("rank %d: no error although rank 0 has no id" % rank)
except RuntimeError as e:
    assert "rank 0" in str(e)
# Two-phase set-up (the advisor's partial-failure case): build_native() is purely local, connect() is the collective part
# (here: a real broadcast, which would block for ever if only one rank entered it).  When the local part fails on ONE
# rank, NO rank enters connect(); when it works everywhere, every rank does.
entered = []
def local_native():
    if rank == fail_on:
        raise RuntimeError("no RCCL library can be bound in this process")
    return Obj("native2")
def connect(o):
    entered.append(o.kind)
    box = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)        # the id hand-off of NativeComm
    assert box[0] == bytes(range(128))
obj2, which2 = agree_on_transport(local_native, fallback, world, all_reduce_min, log=logs.append, connect=connect)
assert which2 == want, (rank, which2)
assert entered == (["native2"] if fail_on < 0 else []), (rank, entered)
# ... and a collective part that fails on every rank alike (rank 0 has no id) also ends in the fallback everywhere
def connect_fails(o):
    NativeComm(FakeLib(True), rank, world)
obj3, which3 = agree_on_transport(lambda: Obj("native3"), fallback, world, all_reduce_min, log=logs.append, connect=connect_fails)
assert which3 == "fallback" and obj3.kind == "torch"
# every rank ended on the same transport
kinds = [None] * world
dist.all_gather_object(kinds, obj.kind)
assert len(set(kinds)) == 1, kinds
# the time split bench.py's frame mode uses at every N the driver runs
for w in (1, 2, 4, 8):
    spans = [shard_range(8192, r, w) for r in range(w)]
    assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert spans[-1][0] + spans[-1][1] == 8192 and all(n % 16 == 0 for _, n in spans)
# the frequency-sliced form (reduce-scatter + sliced merge + all-gather) over THIS world size: every rank owns one slice of the
# flattened [bin][x] arrays, the slices tile the state exactly, and "sum everywhere, keep your slice, gather the slices" gives every
# rank the sum of all ranks' counts (integer: order-independent, display.cl:161-177) -- for the geometries the bench shards
from gr_fosphor_amd.dist import slice_range
for cells in (1024 * 256, 8192 * 512, 65536 * 512):
    spans = [slice_range(cells, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == cells and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert len(set(b - a for a, b in spans)) == 1
cells = 1024 * 64
gen = torch.Generator().manual_seed(1000 + rank)
mine = torch.randint(0, 1000, (cells,), dtype=torch.int32, generator=gen)
want = torch.zeros(cells, dtype=torch.int32)
for r in range(world):
    want += torch.randint(0, 1000, (cells,), dtype=torch.int32, generator=torch.Generator().manual_seed(1000 + r))
summed = mine.clone()
dist.all_reduce(summed, op=dist.ReduceOp.SUM)
lo, hi = slice_range(cells, rank, world)
own = torch.full((cells,), -1, dtype=torch.int32)
own[lo:hi] = summed[lo:hi]                      # what a rank holds after the reduce-scatter: its slice only
pieces = [torch.empty(hi - lo, dtype=torch.int32) for _ in range(world)]
dist.all_gather(pieces, own[lo:hi].contiguous())
assert torch.equal(torch.cat(pieces), want), "rank %d: gathered slices are not the sum of all ranks' counts" % rank
try:
    slice_range(1001, 0, world)
    raise SystemExit("slice_range accepted a ragged split")
except ValueError:
    pass
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


@pytest.mark.parametrize("world,fail_on", [(2, -1), (2, 0), (2, 1), (8, -1), (8, 5)])
def test_transport_agreement_gloo(tmp_path, world, fail_on):
    """bench.py --gpus N / ShardedFosphor: if the library's own RCCL communicator cannot be set up on ANY rank, EVERY rank
    falls back to the torch transport (and a rank whose set-up had succeeded closes it again); the time split (shard_range) and
    the frequency-sliced cell ranges (slice_range) tile the frame / the state at the world size under test.  World size 8 is the
    driver's largest run: the first 8-rank contact of this logic must not be on the GPU node."""
    script = tmp_path / "worker.py"
    script.write_text(TRANSPORT_WORKER)
    env = dict(os.environ, FOSPHOR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29631 + 10 * world + fail_on),
               WORLD_SIZE=str(world), OMP_NUM_THREADS="1", FAIL_ON=str(fail_on))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
        assert "rank %d ok" % r in out
