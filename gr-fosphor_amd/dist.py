"""Multi-GPU frame: one process per GPU, time-sharded batch, one RCCL exchange per display frame.

SURVEY 8e: the spectra of one batch are independent through FFT, log-power and binning; all
cross-spectrum coupling is a commutative reduction per column --
    hit counts  hc[bin][x]   uint32, SUM   (order-independent -> bit-exact on any ring/tree)
    live sum    S[x]         float,  SUM   (weights use the GLOBAL time index)
    max         M[x]         float,  MAX
after which every rank applies the identical merge kernel (K3) to identical inputs, so the
persistent state is replicated bit-identically.  Waterfall rows stay with the rank that
computed them.

The exchange itself is native: `fosphor_amd_exchange` (gr-fosphor_amd/csrc/fosphor_exchange.cpp)
issues ONE ncclGroup of three all-reduces on the library's count/merge stream, between K2 and K3,
so a frame costs the host three C calls and the exchange of frame k overlaps K1 of frame k + 1.
The communicator is the library's own (`NativeComm`); torch.distributed is only used to hand the
128-byte RCCL id from rank 0 to the others.  For large states (65536 x 512: 128 MiB of counts) the
frequency-sliced form reduce-scatters the counts and lets every rank merge only its slice.

`allreduce_partials` / `combine_partials_numpy` are the same combination rule on torch / numpy
arrays: the CPU tests (gloo, world size 2) and two-ranks-on-one-GPU tests use them as transport.
"""
import ctypes as C

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
    """The combination rule through torch.distributed: three all-reduces, in place, on any backend (gloo on
    CPU tensors in the tests).  hc must be an integer tensor (uint32 counts viewed as int32: sums stay below
    2^31 for any batch < 2^31 spectra).  async_op=True returns the work handles."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
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


def slice_range(cells, rank, world):
    """Frequency-sliced merge (fosphor_amd_exchange_sliced / fosphor_amd_merge_sliced): rank r owns cells
    [r C / world, (r + 1) C / world) of the flattened [bin][x] count / histogram arrays.  ValueError when the state does not
    split evenly (ShardedFosphor then uses the all-reduce form)."""
    if cells % world:
        raise ValueError("%d cells do not split into %d slices" % (cells, world))
    per = cells // world
    return per * rank, per * (rank + 1)


class NativeComm:
    """The library's own RCCL communicator (fosphor_amd_comm_*).  `broadcast_id(id_bytes_or_None) -> bytes`
    hands rank 0's 128-byte id to every rank; by default torch.distributed does it (any backend)."""

    @staticmethod
    def available(lib):
        """True when the RCCL library can be bound in this process.  Purely local: no collective, no communicator."""
        return int(lib.fosphor_amd_comm_available()) == 1

    def __init__(self, lib, rank, world, broadcast_id=None):
        self.L, self.rank, self.world = lib, rank, world
        buf = (C.c_char * 128)()
        rv0 = lib.fosphor_amd_comm_unique_id(buf) if rank == 0 else 0
        ident = bytes(buf.raw) if not rv0 else b""	# an empty id tells the other ranks that rank 0 has none: nobody is left waiting
        if world > 1:
            ident = (broadcast_id or self._torch_broadcast)(ident if rank == 0 else None)
        if not ident:
            raise RuntimeError("fosphor_amd_comm_unique_id failed on rank 0 (%d): is an RCCL library available?" % rv0)
        self.h = C.c_void_p()
        rv = lib.fosphor_amd_comm_init(C.byref(self.h), world, rank, ident)
        if rv:
            raise RuntimeError("fosphor_amd_comm_init -> %d" % rv)

    @staticmethod
    def _torch_broadcast(ident):
        import torch.distributed as dist
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def count(self):
        """ncclCommCount of the communicator: how many ranks RCCL itself says it spans."""
        return int(self.L.fosphor_amd_comm_count(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.fosphor_amd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedFosphor:
    """One rank's part of a sharded fosphor instance.

    frame(d_samples_local, total_batch): K1 + K2 on the local time block, exchange, K3.
    The semantics are one reference display launch with fft_batch = total_batch (the kernel
    is batch-generic; only the host caps it, cl.c:885).

    exchange = "rccl"   (default on GPUs) the native exchange on the library's count/merge stream: nothing
                        waits on the host, frame k's exchange overlaps frame k+1's K1 by construction.
               "torch"  the same three all-reduces through torch.distributed on views of the library's arrays
                        (any backend; what the two-ranks-on-one-GPU gloo test uses).  With overlap=True the
                        exchange of frame k is left in flight while frame k+1 is submitted (two slots).
    sliced: reduce-scatter + frequency-sliced merge (rccl only); default for states of 16 MiB and more.
    force_exchange: run the collectives on a single rank too (smoke test of the RCCL path).
    connect=False: do everything that can fail on this rank ALONE (bind RCCL, construct the instance) and leave the
                collective part -- rank 0's id to every rank, ncclCommInitRank -- to connect(): agree_on_transport puts
                an agreement between the two, so that a rank whose local part failed leaves nobody waiting in a collective.
    """

    def __init__(self, fosphor_cls, rank, world, group=None, exchange=None, sliced=None, force_exchange=False,
                 comm=None, connect=True, **kw):
        import os
        import torch
        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        # K1 runs on a torch-owned, NON-default stream (torch's default stream has handle 0, which the C ABI
        # reads as "create a private stream"), so that the caller's producer of the samples can be ordered
        # in front of it with wait_stream.
        self.stream = torch.cuda.Stream()
        self.f = fosphor_cls(stream=self.stream.cuda_stream, **kw)
        self.force = bool(force_exchange or os.environ.get("FOSPHOR_AMD_FORCE_EXCHANGE"))
        if exchange is None:
            exchange = os.environ.get("FOSPHOR_AMD_EXCHANGE", "rccl")
        self.exchange = exchange
        self.active = world > 1 or self.force
        cells = self.f.n_bins * self.f.n
        self.sliced = (cells * 4 >= (16 << 20)) if sliced is None else bool(sliced)
        if cells % world:
            self.sliced = False
        self.comm = comm
        if self.active and exchange == "rccl":
            if comm is None and not NativeComm.available(self.f.L):
                self.f.close()
                raise RuntimeError("no RCCL library can be bound in this process")
            if comm is None and connect:
                self.connect()
        else:
            self.sliced = False
        # "torch" transport: K2 / all-reduce / K3 live on the library's second stream
        self.stream_b = torch.cuda.ExternalStream(self.f.stream2)
        self.views = []
        if self.active and exchange == "torch":
            for slot in (0, 1):
                self.f.set_partial_slot(slot)
                p = self.f.partials()
                self.views.append((wrap_device_array(p.d_hc, (p.n_hc,), torch.int32),
                                   wrap_device_array(p.d_live_sum, (p.n_cols,), torch.float32),
                                   wrap_device_array(p.d_max, (p.n_cols,), torch.float32)))
        self.f.set_partial_slot(0)
        self.k = 0
        self.pending = None		# torch transport: (works, slot, total_batch)

    def connect(self, broadcast_id=None):
        """The collective part of the native transport (every rank calls it, or none)."""
        if self.active and self.exchange == "rccl" and self.comm is None:
            self.comm = NativeComm(self.f.L, self.rank, self.world, broadcast_id)

    def exchange_ranks(self):
        """Ranks the exchange spans, as the transport itself reports it (ncclCommCount / the process group's size)."""
        if not self.active:
            return 1
        if self.comm is not None:
            return self.comm.count()
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    # ---- torch transport --------------------------------------------------------
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

    def frame(self, d_samples_local, total_batch, overlap=False, wait_producer=True, overlap_ratio=1):
        """wait_producer=False: the caller guarantees d_samples_local is complete (saves the event
        record + wait between the caller's stream and the FFT stream, two queue packets per frame).
        overlap_ratio > 1: d_samples_local is this rank's part of the unexpanded stream (overlap_cc fused into the read)."""
        torch = self.torch
        off, n = shard_range(total_batch, self.rank, self.world)
        if wait_producer:
            self.stream.wait_stream(torch.cuda.current_stream())	# the caller's producer of d_samples_local
        if self.exchange == "torch" and self.active:
            slot = self.k & 1
            self.k += 1
            with torch.cuda.stream(self.stream):
                self.f.set_partial_slot(slot)
                rv = self.f.accumulate_device(d_samples_local, n, off, total_batch, overlap_ratio)	# K1 here, K2 on stream_b
                if rv:
                    raise RuntimeError("accumulate_device -> %d" % rv)
            with torch.cuda.stream(self.stream_b):
                import torch.distributed as dist
                h, s, m = self.views[slot]
                works = [dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                         dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                         dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group, async_op=True)]
                self._retire()			# previous frame: wait for ITS exchange, merge (K3 on stream_b)
                self.pending = (works, slot, total_batch)
                if not overlap:
                    self._retire()
            return
        # native transport (or a single rank without exchange): three C calls, all asynchronous; K2, the
        # exchange and K3 follow each other on the library's count/merge stream while `stream` is already free
        # for the next frame's K1
        rv = self.f.accumulate_device(d_samples_local, n, off, total_batch, overlap_ratio)
        if rv:
            raise RuntimeError("accumulate_device -> %d" % rv)
        if self.comm is not None:
            if self.sliced:
                rv = self.f.exchange_sliced(self.comm.h, self.world, self.rank)
                rv = rv or self.f.merge_sliced(total_batch, self.world, self.rank)
            else:
                rv = self.f.exchange(self.comm.h)
                rv = rv or self.f.merge(total_batch)
        else:
            rv = self.f.merge(total_batch)
        if rv:
            raise RuntimeError("exchange / merge -> %d" % rv)

    def gather_state(self):
        """Sliced mode: make the histogram complete on every rank (once per draw)."""
        if self.sliced and self.comm is not None:
            rv = self.f.gather_state(self.comm.h, self.world, self.rank)
            if rv:
                raise RuntimeError("gather_state -> %d" % rv)

    def flush(self):
        if self.pending is not None:
            with self.torch.cuda.stream(self.stream_b):
                self._retire()

    def close(self):
        self.flush()
        self.f.finish()
        if self.comm is not None:
            self.comm.close()
        self.f.close()


def agree_on_transport(build_native, build_fallback, world, all_reduce_min, log=None, connect=None):
    """Every rank must exchange through the same transport.

    Two phases when `connect` is given (what bench.py and ShardedFosphor(connect=False) use):
      1. build_native() does ONLY what can fail on one rank alone -- bind the RCCL library, construct the instance --
         and contains no collective; one MIN all-reduce of "it worked here" follows.  A rank whose RCCL cannot be
         loaded, or whose instance fails to initialise, therefore never leaves the others waiting in the id broadcast
         or in ncclCommInitRank: unless phase 1 worked everywhere, every rank builds build_fallback() instead.
      2. connect(obj) is the collective part (rank 0's id to every rank, ncclCommInitRank), entered by every rank or
         by none; a second MIN all-reduce catches a failure that every rank sees (rank 0 has no id, RCCL refuses the
         topology).  A rank that dies INSIDE the collective can still stall the others: that is RCCL's own failure
         mode and is left to its time-outs (NCCL_TIMEOUT / the launcher's).
    Without `connect`, build_native() is the whole set-up (it may contain collectives; only failures that happen before
    them, or on every rank alike, are caught).
    all_reduce_min(int) -> int is the caller's collective (torch.distributed on any backend).  Nothing here touches a GPU or
    re-executes anything: it can run before or after the first HIP call.  Returns (object, "native" | "fallback")."""
    def agreed(flag):
        return int(all_reduce_min(flag)) if world > 1 else flag

    try:
        obj, ok = build_native(), 1
    except Exception as e:
        if log:
            log("native exchange unavailable on this rank (%s)" % e)
        obj, ok = None, 0
    ok = agreed(ok)
    if ok and connect is not None:
        try:
            connect(obj)
        except Exception as e:
            if log:
                log("native communicator could not be set up (%s)" % e)
            ok = 0
        ok = agreed(ok)
    if ok:
        return obj, "native"
    if obj is not None:
        obj.close()
    return build_fallback(), "fallback"


def combine_partials_numpy(parts):
    """Reference combination rule on host arrays: [(hc, live, max), ...] -> (hc, live, max)."""
    hc = np.sum([p[0].astype(np.uint64) for p in parts], axis=0).astype(np.uint32)
    live = np.sum([p[1].astype(np.float32) for p in parts], axis=0, dtype=np.float32)
    vmax = np.max([p[2] for p in parts], axis=0)
    return hc, live, vmax
