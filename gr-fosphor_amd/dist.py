"""Multi-GPU frame: one process per GPU, time-sharded batch, per-frame RCCL all-reduce.

SURVEY 8e: the spectra of one batch are independent through FFT, log-power and binning; all
cross-spectrum coupling is a commutative reduction per column --
    hit counts  hc[bin][x]   uint32, SUM   (order-independent -> bit-exact on any ring/tree)
    live sum    S[x]         float,  SUM   (weights use the GLOBAL time index)
    max         M[x]         float,  MAX
after which every rank applies the identical merge kernel (K3) to identical inputs, so the
persistent state is replicated bit-identically.  Waterfall rows stay with the rank that
computed them.  torch.distributed ("nccl" = RCCL on ROCm) is the transport; the payload at
1024 x 128 is 0.52 MiB, i.e. latency-bound on xGMI, so the three arrays go out as one
coalesced group per frame.
"""
import numpy as np


class _DeviceArray:
    """Minimal __cuda_array_interface__ holder so torch can view library-owned HBM."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
        }


_TYPESTR = {"torch.int32": "<i4", "torch.float32": "<f4", "torch.uint8": "|u1"}


def wrap_device_array(ptr, shape, dtype):
    """torch tensor aliasing `ptr` (no copy).  dtype: torch.int32 / float32 / uint8."""
    import torch
    return torch.as_tensor(_DeviceArray(ptr, shape, _TYPESTR[str(dtype)]), device="cuda")


def allreduce_partials(hc, live_sum, vmax, group=None, async_op=False):
    """The per-frame exchange.  Tensors are reduced in place; works on any backend
    (RCCL on GPU tensors, gloo on CPU tensors in the tests).  hc must be an integer tensor
    (uint32 counts viewed as int32: sums stay below 2^31 for any batch < 2^31 spectra).
    async_op=True returns the work handles (wait() makes the current stream wait, not the host)."""
    import os
    import torch.distributed as dist
    if not dist.is_initialized():
        return []
    if dist.get_world_size(group) == 1 and not os.environ.get("FOSPHOR_AMD_FORCE_EXCHANGE"):
        return []		# (the override runs the collectives on a single rank: a smoke test of the RCCL path)
    works = [
        dist.all_reduce(hc, op=dist.ReduceOp.SUM, group=group, async_op=True),
        dist.all_reduce(live_sum, op=dist.ReduceOp.SUM, group=group, async_op=True),
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX, group=group, async_op=True),
    ]
    if async_op:
        return works
    for w in works:
        w.wait()
    return []


def shard_range(total_batch, rank, world):
    """Contiguous time block of a batch for one rank (each a multiple of 16 spectra)."""
    if total_batch % (16 * world):
        raise ValueError("batch %d does not split into %d shards of whole 16-spectrum groups" % (total_batch, world))
    n = total_batch // world
    return rank * n, n


class ShardedFosphor:
    """One rank's half of a sharded fosphor instance.

    frame(d_samples_local, total_batch): K1+K2 on the local shard, all-reduce, K3.
    The semantics are one reference display launch with fft_batch = total_batch (the kernel
    is batch-generic; only the host caps it, cl.c:885).

    With overlap=True the exchange of frame k is left in flight while the caller submits
    frame k+1 (two partial-array slots in the library); its merge is queued behind frame k+1's
    FFT.  Call flush() to retire the last frame.
    """

    def __init__(self, fosphor_cls, rank, world, group=None, **kw):
        import torch
        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        # The library runs on a torch-owned, NON-default stream and every frame is submitted under
        # it, so the collective (which torch orders after the current stream) follows K2, and K3
        # follows the collective (work.wait() makes this stream wait) without host synchronisation.
        # torch's default stream has handle 0, which the C ABI reads as "create a private stream":
        # that would silently break the ordering, hence the explicit stream.
        self.stream = torch.cuda.Stream()
        self.f = fosphor_cls(stream=self.stream.cuda_stream, **kw)
        # K2 / all-reduce / K3 live on the library's second stream, so that they overlap the
        # NEXT frame's K1 (VALU-bound) instead of queueing behind it
        self.stream_b = torch.cuda.ExternalStream(self.f.stream2)
        self.views = []
        for slot in (0, 1):
            self.f.set_partial_slot(slot)
            p = self.f.partials()
            self.views.append((wrap_device_array(p.d_hc, (p.n_hc,), torch.int32),
                               wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32),
                               wrap_device_array(p.d_max, (p.n_cols,), torch.float32)))
        self.k = 0
        self.pending = None		# (works, slot, total_batch)

    def _retire(self):
        if self.pending is None:
            return
        works, slot, total = self.pending
        for w in works:
            w.wait()
        self.f.set_partial_slot(slot)
        rv = self.f.merge(total)
        if rv:
            raise RuntimeError("merge -> %d" % rv)
        self.pending = None

    def frame(self, d_samples_local, total_batch, overlap=False, wait_producer=True):
        """wait_producer=False: the caller guarantees d_samples_local is complete (saves the event
        record + wait between the caller's stream and the FFT stream, two queue packets per frame)."""
        torch = self.torch
        off, n = shard_range(total_batch, self.rank, self.world)
        slot = self.k & 1
        self.k += 1
        if wait_producer:
            self.stream.wait_stream(torch.cuda.current_stream())	# the caller's producer of d_samples_local
        with torch.cuda.stream(self.stream):
            self.f.set_partial_slot(slot)
            rv = self.f.accumulate_device(d_samples_local, n, off, total_batch)	# K1 here, K2 on stream_b
            if rv:
                raise RuntimeError("accumulate_device -> %d" % rv)
        with torch.cuda.stream(self.stream_b):
            works = allreduce_partials(*self.views[slot], group=self.group, async_op=True)
            self._retire()			# previous frame: wait for ITS exchange, merge (K3 on stream_b)
            self.pending = (works, slot, total_batch)
            if not overlap:
                self._retire()

    def flush(self):
        with self.torch.cuda.stream(self.stream_b):
            self._retire()


def combine_partials_numpy(parts):
    """Reference combination rule on host arrays: [(hc, live, max), ...] -> (hc, live, max)."""
    hc = np.sum([p[0].astype(np.uint64) for p in parts], axis=0).astype(np.uint32)
    live = np.sum([p[1].astype(np.float32) for p in parts], axis=0, dtype=np.float32)
    vmax = np.max([p[2] for p in parts], axis=0)
    return hc, live, vmax
