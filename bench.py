#!/usr/bin/env python3
"""Headline benchmark: complex IQ MSamples/s through the fosphor hot path at 1024-pt FFT.

Contract (one JSON line on rank 0):
    python bench.py --gpus N --steps K --warmup W
    N > 1 without a launcher (WORLD_SIZE unset): this process starts
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
    as a CHILD, before anything here has touched the GPU, relays rank 0's JSON line and exits with the child's code.
    Under a launcher (WORLD_SIZE set) it is one rank of N.

A STEP is ONE CALL of the library's device-resident entry point with --batches-per-step (256)
reference batches of BASELINE config C2 -- 1024 spectra x 1024 points each, 1024x256 histogram +
waterfall -- i.e. 268 435 456 complex samples = 2 GiB of fp32 IQ already resident in HBM, taken
through the whole path: windowed FFT -> log-power -> exact histogram bin -> hit counts ->
persistence-histogram rise/decay, live EMA, max-hold, waterfall.  The library cuts a call into
sub-launches of 64 batches (K1 | K2 | K3 pipelined over streams); EVERY batch gets its own state
update in order, exactly as 256 successive fosphor_process() calls of the reference would give
(cl.c:870-968; tests/test_gpu_parity.py::test_multi_batch_launch_equals_sequential_calls).
--steps / --warmup count such calls.  Before the warm-up an UNTIMED pre-conditioning phase
(--precondition seconds, default 0.5 s, reported) runs the same steps so that the clocks have
settled: the first milliseconds of a run are 10-30 % slower.

  N = 1  ("batch" mode): as above.
  N > 1  ("frame" mode): the spectra of a display frame (256 steps-worth of batches per GPU) are
         time-sharded over the ranks; hit counts / live sums / max are all-reduced over RCCL once per
         frame and every rank applies the same state update; per-GPU work is the same at
         every N (weak scaling, BASELINE configs[3] "C4").  FOSPHOR_AMD_FORCE_EXCHANGE=1 runs the
         collectives on a single rank.  `--mode frame --batches-per-step 1` is SURVEY 8e's "K = 1, honest
         worst case": one exchange per 1024-spectrum launch per GPU instead of one per display frame.
         A multi-rank line is a scaling record only if the exchange was the native RCCL one over all N ranks
         (config.transport, config.exchange_ranks_per_rank) and left right results (config.exchange_check, after the
         timed region: every spectrum of the last frame counted once per column on every rank, the replicated state
         bit-identical across ranks); otherwise the line is marked `invalid` and the run exits 3.

Other BASELINE configurations: --config C3 (8192-pt FFT, 50 % overlap fused into the read, batch
4096, 512 bins) and --config C5 (65536-pt FFT, fp16 IQ, 512 bins, the per-GPU share of a sharded
frame) emit the same JSON shape with their own workload string and roofline convention.  The default
run (C2, one GPU) ALSO measures C3 (20 steps) and C5 (200 steps: a C5 step is one 0.29 ms frame), in the same process, after the
headline, and reports them inside the one JSON line as `other_configs` (--no-other-configs skips them), followed by
`other_configs.sink`: the PCIe-INCLUSIVE rates of the drop-in path (work() -> FIFO -> H2D -> kernels, and plain fosphor_process()
calls of 1 Mi host samples), 2 s each, as MSamples/s and as a fraction of the Gen5 x16 link -- reported, never `value`.

roofline: the dominant kernel is K1 (fft_bin), bound by the HBM read of the IQ stream (8 B per
sample, 4 B for fp16 IQ).  K1s of consecutive sub-launches run on alternating streams and overlap
at their edges, so `achieved` = algorithmic bytes of all K1 launches / time during which at least
one K1 was running (union of the hipEvent intervals recorded on the streams the kernels run on);
the plain per-launch average (which counts shared time twice) is reported beside it.
cpu_baseline: the oracle (oracle/fosphor_oracle.c, the CPU restatement of the reference's
fft.cl + display.cl) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0		# MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
ACHIEVABLE_GBS = 6290.0		# measured float4 copy on MI355X (guide), informational

# name -> geometry.  spb = spectra per batch, bps = default batches per step, over = overlap_cc ratio
CONFIGS = {
    "C2": dict(log2n=10, bins=256, spb=1024, bps=256, over=1, fp16=False,
               text="C2: 1024-pt FFT, batch=1024 spectra, 1024x%(bins)d histogram + waterfall; one step = one "
                    "fosphor_amd_process_device call of %(bps)d such batches (%(msamp)d Mi samples, %(mib)d MiB of IQ), "
                    "sub-launched %(sub)d batches at a time, every batch with its own state update"),
    "C3": dict(log2n=13, bins=512, spb=4096, bps=28, over=2, fp16=False,
               text="C3: 8192-pt FFT, 50 %% overlap (overlap_cc(8192, 2) fused into the read), batch=4096 spectra, "
                    "8192x%(bins)d histogram + waterfall; one step = one fosphor_amd_process_device_overlap call of "
                    "%(bps)d such batches (%(msamp)d Mi FFT'd samples)"),
    "C5": dict(log2n=16, bins=512, spb=1024, bps=1, over=1, fp16=True,
               text="C5: 65536-pt FFT, fp16 IQ, 65536x%(bins)d histogram + waterfall; one step = one display frame of 1024 "
                    "spectra (what the 8 GPUs of BASELINE configs[4] share 128 spectra each) as one batch on this GPU: one "
                    "fosphor_amd_process_device call of %(bps)d such batch (%(msamp)d Mi samples), one state update"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--bins", type=int, default=0)
    ap.add_argument("--batches-per-step", type=int, default=0, help="reference batches per step (one library call)")
    ap.add_argument("--ring-steps", type=int, default=2, help="distinct steps of IQ resident in HBM")
    ap.add_argument("--precondition", type=float, default=0.5, help="seconds of untimed steps before the warm-up")
    ap.add_argument("--mode", choices=["auto", "batch", "frame"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic-twin", action="store_true")
    ap.add_argument("--no-placement-tuning", action="store_true",
                    help="take the allocations as they come (default: fosphor_amd_tune_placement once, untimed, before the pre-conditioning)")
    ap.add_argument("--placement-candidates", type=int, default=1,
                    help="batch mode, NOT for headline runs: instances (= sets of allocations) tried before the run; the fastest over 48 untimed steps is kept")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the informational K2/K3 and isolated-K1 passes")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default C2 run on one GPU: do not measure C3 / C5 afterwards (other_configs in the JSON line)")
    ap.add_argument("--other-steps", type=int, default=20, help="steps of the other_configs pass (C3; C5 runs ten times as many)")
    ap.add_argument("--sink-seconds", type=float, default=2.0, help="seconds of each PCIe-inclusive leg of other_configs.sink")
    ap.add_argument("--strict-ordering", action="store_true", help="keep stream ordering between calls (default: relaxed, "
                    "the input ring is never rewritten)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start one as a child (no GPU call has been made in this process)
# ---------------------------------------------------------------------------------------------------

def scaling_record_guard(world, transport, exchange_ranks_per_rank, test_hook=False, check=None):
    """A line measured on N > 1 ranks is a SCALING record (the driver computes efficiency from it): it must have exchanged through
    the native RCCL transport, and the library's communicator must span all N ranks on every rank (ncclCommCount).  Returns the
    reason the record is invalid, or None.  check: config.exchange_check of the line (the reduced hit counts hold every spectrum of the
    last frame once per column on every rank; the replicated state is bit-identical across ranks) -- wrong results fail the run
    whatever the transport.  test_hook: the ranks share one GPU over gloo on purpose (tests/test_gpu_dist.py): the torch transport
    is then what is being tested."""
    if check is not None and world > 1 and not (check.get("every_spectrum_counted_once_on_every_rank")
                                                and check.get("replicated_state_bit_identical_across_ranks")):
        return "the exchange left wrong results (%s): not a scaling record" % check
    if world <= 1 or test_hook:
        return None
    if not str(transport).startswith("native RCCL"):
        return "%d ranks exchanged through '%s', not the native RCCL transport: not a scaling record" % (world, transport)
    bad = [(r, n) for r, n in enumerate(exchange_ranks_per_rank) if int(n) != world]
    if len(exchange_ranks_per_rank) != world or bad:
        return "the exchange spans %s ranks per rank, not %d on every rank: not a scaling record" % (list(exchange_ranks_per_rank), world)
    return None


def launcher_command(n_gpus, argv, port):
    """The command `bench.py --gpus N` starts when no launcher did (pure function: tests/test_boundary_cpu.py)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """Child process, output relayed: rank 0's JSON line goes to our stdout (once), everything else to stderr; our exit code is the
    child's.  Nothing here imports torch or touches the GPU -- a process that has initialised the GPU must never be replaced."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")		# dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launcher_command(args.gpus, sys.argv[1:], port)
    sys.stderr.write("bench.py: no launcher (WORLD_SIZE unset), starting: %s\n" % " ".join(cmd))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line_seen = False
    for line in proc.stdout:
        st = line.strip()
        if not line_seen and st.startswith("{") and '"metric"' in st:
            print(st, flush=True)
            line_seen = True
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and not line_seen:
        sys.stderr.write("bench.py: the launched ranks exited 0 without a JSON line\n")
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------

def numa_cpu_groups(allowed):
    """[[cpu, ...] per NUMA node] restricted to the CPUs this process may run on (one group when the topology cannot be read)."""
    groups = []
    try:
        base = "/sys/devices/system/node"
        for d in sorted((x for x in os.listdir(base) if x.startswith("node") and x[4:].isdigit()), key=lambda x: int(x[4:])):
            cpus = []
            for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                if not part:
                    continue
                a, _, b = part.partition("-")
                cpus.extend(range(int(a), int(b or a) + 1))
            cpus = [c for c in cpus if c in allowed]
            if cpus:
                groups.append(cpus)
    except Exception:
        groups = []
    if not groups:
        groups = [sorted(allowed)]
    return groups


def cpu_baseline(bins, seconds):
    """Oracle (CPU restatement of the reference kernels), bounded sample, the best the host does.

    One oracle instance parallelises a batch over at most 64 threads (its display stage has 64 column groups of 16,
    cl.c:945-950).  Two legs, half the time budget each:
      all-core    one instance per 64 CPUs of every NUMA node, each PINNED to its CPUs (sched_setaffinity in the worker thread,
                  inherited by the oracle's pthreads) with its input and its state first-touched from there -- the way several
                  sink blocks would share the host;
      one-instance  a single instance of 64 threads pinned to (the first 64 CPUs of) one node.
    value = the faster leg; cores = the threads that leg actually used.  The instrument this stands beside: main.c:141-155."""
    import threading
    import numpy as np
    from oracle_lib import Oracle, build_oracle, gaussian_iq
    build_oracle(ref=False)
    allowed = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    nodes = numa_cpu_groups(allowed)
    x0 = gaussian_iq(1024 * 1024, 7)

    def leg(groups, budget):
        counts = [0] * len(groups)
        ready = threading.Barrier(len(groups) + 1)
        t_start = [0.0]

        def work(i):				# (ctypes releases the GIL for the duration of the C call)
            try:
                os.sched_setaffinity(0, groups[i])		# pid 0 = the calling thread; the oracle's pthreads inherit it
            except Exception:
                pass
            x = np.array(x0, copy=True)			# first touch on this node
            o = Oracle(n_bins=bins)
            o.process(x, nthreads=len(groups[i]))		# warm-up: pages in the instance's state, here
            ready.wait()
            while time.perf_counter() - t_start[0] < budget and counts[i] < 4096:
                o.process(x, nthreads=len(groups[i]))
                counts[i] += 1

        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(groups))]
        for t in ths:
            t.start()
        t_start[0] = time.perf_counter() + 1e9		# (workers do not start timing before the barrier releases)
        ready.wait()
        t_start[0] = time.perf_counter()
        for t in ths:
            t.join()
        el = time.perf_counter() - t_start[0]
        n = sum(counts)
        return {"value": n * 1024 * 1024 / el / 1e6, "batches": n, "seconds": el, "instances": len(groups),
                "threads": sum(len(g) for g in groups)}

    all_groups = [node[i:i + 64] for node in nodes for i in range(0, len(node), 64)]
    flat = [c for node in nodes for c in node]
    one_group = [nodes[0][:64] if len(nodes[0]) >= 64 else flat[:64]]
    two = len(all_groups) > 1 or len(all_groups[0]) != len(one_group[0])
    legs = {"all_core": leg(all_groups, seconds / 2 if two else seconds)}
    if two:
        legs["one_instance"] = leg(one_group, seconds / 2)
    best = max(legs, key=lambda k: legs[k]["value"])
    b = legs[best]
    return {"value": b["value"], "unit": "MSamples/s", "cores": b["threads"], "kind": "port",
            "cores_online": len(allowed), "numa_nodes": len(nodes), "leg": best,
            "legs": {k: {"value": v["value"], "cores": v["threads"], "instances": v["instances"]} for k, v in legs.items()},
            "sample": "%d batches of 1024 x 1024-pt spectra (%.1f s), oracle C restatement of fft.cl+display.cl, "
                      "%d instance(s) pinned to their NUMA node's CPUs, %d threads of %d online host cores (%s); the reference's own "
                      "OpenCL path could not be run: no OpenCL CPU runtime (POCL) exists in this image and reference sources do "
                      "not travel to the GPU box" % (b["batches"], b["seconds"], b["instances"], b["threads"], len(allowed),
                                                     "; ".join("%s %.0f MS/s on %d" % (k, v["value"], v["threads"]) for k, v in legs.items()))}


# ---------------------------------------------------------------------------------------------------
# one configuration
# ---------------------------------------------------------------------------------------------------

def measure(name, args, ctx, steps, warmup, light=False, batches_per_step=0):
    """Measures configuration `name`; returns the JSON object (rank 0) or None (other ranks).  light: the other_configs pass --
    no placement tuning, no extra passes, no twin, no CPU leg (those describe the headline)."""
    torch, dist, gr_fosphor_amd = ctx["torch"], ctx["dist"], ctx["pkg"]
    world, rank = ctx["world"], ctx["rank"]
    from gr_fosphor_amd.dist import ShardedFosphor

    cfg = CONFIGS[name]
    n_fft = 1 << cfg["log2n"]
    bins = (args.bins if not light else 0) or cfg["bins"]
    spb = cfg["spb"]
    over = cfg["over"]
    bytes_per_sample = 4 if cfg["fp16"] else 8	# algorithmic read per FFT'd sample (materialised-stream convention)
    mode = args.mode if args.mode != "auto" else ("batch" if world == 1 else "frame")
    if light:
        mode = "batch"
    F = batches_per_step if batches_per_step > 0 else (args.batches_per_step if (args.batches_per_step > 0 and not light) else cfg["bps"])
    ring = max(1, args.ring_steps)
    samples_per_batch = spb * n_fft			# FFT'd samples
    hop = n_fft // over
    # unexpanded stream of one step: (F*spb - 1) * hop + n_fft samples (overlap_cc_impl.cc:64-79)
    step_stream = (F * spb - 1) * hop + n_fft if over > 1 else F * samples_per_batch
    extra = not (args.no_extra_passes or light)
    precondition = args.precondition if not light else min(args.precondition, 0.25)

    # synthetic white complex Gaussian IQ, sigma 0.05 per component, resident in HBM;
    # the ring is larger than the 256 MiB Infinity Cache, so IQ reads come from HBM
    g = torch.Generator(device="cuda")
    g.manual_seed(7 + rank)
    iq = torch.empty((ring * step_stream, 2), dtype=torch.float32, device="cuda")
    iq.normal_(0.0, 0.05, generator=g)
    if cfg["fp16"]:
        iq = iq.to(torch.float16)
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream
    kw = dict(n_bins=bins, max_spectra=F * spb, max_batches=F, fft_len_log=cfg["log2n"], iq_fp16=cfg["fp16"])
    transport = None
    if mode == "batch":
        f = gr_fosphor_amd.Fosphor(stream=stream, **kw)
        sf = None
        if not args.strict_ordering:
            f.set_input_ordering(False)		# the ring is written once, before the first call
    else:
        from gr_fosphor_amd.dist import agree_on_transport

        def all_reduce_min(v):
            t_ok = torch.tensor([v], dtype=torch.int32, device="cuda")
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            return int(t_ok.item())

        # the library's own RCCL communicator, or -- on every rank together -- torch.distributed's all-reduces
        # (two phases: what can fail on one rank alone -- binding RCCL, the instance -- is agreed on before any rank enters the
        # collective part, the id hand-off and ncclCommInitRank)
        sf, transport = agree_on_transport(lambda: ShardedFosphor(gr_fosphor_amd.Fosphor, rank, world, connect=False, **kw),
                                           lambda: ShardedFosphor(gr_fosphor_amd.Fosphor, rank, world, exchange="torch", **kw),
                                           world, all_reduce_min,
                                           log=lambda m: sys.stderr.write("rank %d: %s\n" % (rank, m)),
                                           connect=lambda o: o.connect())
        f = sf.f
        if not args.strict_ordering:
            f.set_input_ordering(False)

    state = {"pos": 0}

    def run_steps(n_steps):
        for _ in range(n_steps):
            pos = state["pos"]
            state["pos"] = (pos + 1) % ring
            view = iq[pos * step_stream:(pos + 1) * step_stream]
            if mode == "batch":
                rv = f.process_device_overlap(view, F, spb, over) if over > 1 else f.process_device(view, F, spb)
                if rv:
                    raise RuntimeError("process_device -> %d" % rv)
            else:
                sf.frame(view, F * spb * world, overlap=True, wait_producer=False, overlap_ratio=over)
        if sf is not None:
            sf.flush()

    def sync():
        if f.finish() < 0:
            raise RuntimeError("device error")
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n):
        t_s = time.perf_counter()
        run_steps(n)
        sync()
        return n * F * samples_per_batch / (time.perf_counter() - t_s) / 1e6

    f.finish()				# instance boot (table uploads, initial fills: cl.c:981-995) is not a step
    # Placement (untimed, once): on MI355X the FFT kernel's memory traffic runs in one of two states -- 98 or 109 us per 512 MiB of IQ --
    # decided by the allocations involved (the IQ buffer against the instance's intermediate sets).  The library
    # re-allocates its sets until the traffic runs at 6 TB/s (fosphor_amd_tune_placement); if none does, the IQ ring is the unlucky side
    # and is allocated again (same distribution, the generator's next numbers).  The rate of the allocations AS THEY CAME is measured
    # first and reported beside the headline (placement.untuned_value).
    placement = None
    if cfg["log2n"] == 10 and bins <= 256 and not args.no_placement_tuning and not light and F >= 16:	# (a launch long enough for its time to be bandwidth)
        sub_t = min(F, 64)
        n_t = sub_t * samples_per_batch
        good_us = (n_t * 9.0 + n_t / 64 * 8.0) / 6.0e12 * 1e6
        placement = {"iq_allocations": 1, "sets_replaced": 0, "twin_us_first": None, "twin_us_final": None, "good_us": good_us}
        if mode == "batch":
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < min(precondition, 0.3):
                run_steps(4); sync()
            placement["untuned_value"] = timed(max(8, min(steps, 40)))	# MSamples/s before any re-allocation (untimed extra pass)
            state["pos"] = 0

        def tune(inst):
            nonlocal iq
            for attempt in range(3):
                r, b_us, a_us = inst.tune_placement(iq[:n_t], sub_t, spb, 6)
                placement["sets_replaced"] += r
                if placement["twin_us_first"] is None:
                    placement["twin_us_first"] = b_us
                placement["twin_us_final"] = a_us
                if a_us <= good_us or attempt == 2 or len(cands) > 0:	# (the ring is re-rolled for the first candidate only)
                    break
                old = iq
                iq = torch.empty_like(old)
                iq.normal_(0.0, 0.05, generator=g)
                torch.cuda.synchronize()
                del old
                placement["iq_allocations"] += 1

        cands = []
        tune(f)
        if mode == "batch" and args.placement_candidates > 1:
            # The twin sees the FFT kernel's own pair of streams; the count and merge kernels have theirs.  What decides is the pipeline:
            # a few candidate instances (each a fresh set of allocations), 48 untimed steps each, the fastest stays.
            def quick_rate():
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < min(precondition, 0.2):	# (every candidate warmed alike: clocks, first touches)
                    run_steps(4); sync()
                return timed(48)
            cands.append((quick_rate(), f))
            for c in range(1, args.placement_candidates):
                f = gr_fosphor_amd.Fosphor(stream=stream, **kw)
                if not args.strict_ordering:
                    f.set_input_ordering(False)
                f.finish()
                tune(f)
                cands.append((quick_rate(), f))
            best = max(range(len(cands)), key=lambda i: cands[i][0])
            placement["candidate_rates"] = [round(c[0]) for c in cands]
            placement["candidate_kept"] = best
            f = cands[best][1]
            for i, c in enumerate(cands):
                if i != best:
                    c[1].close()
            state["pos"] = 0
    # untimed pre-conditioning: the same steps until the clocks have settled
    t0 = time.perf_counter()
    pre_steps = 0
    while time.perf_counter() - t0 < precondition:
        run_steps(4)
        sync()
        pre_steps += 4
    precondition_s = time.perf_counter() - t0

    run_steps(warmup)
    sync()
    share0 = f.share_stats()
    # timed region: hipEvents around K1 only (events around K2/K3 too sit on the critical path of the
    # count/merge stream)
    if not os.environ.get("BENCH_NO_PROFILE"):	# debugging aid: cost of the hipEvents themselves
        f.profile(2)
    t0 = time.perf_counter()
    run_steps(steps)
    t_submit = time.perf_counter() - t0		# host time to queue everything (host-bound if ~ elapsed)
    sync()
    elapsed = time.perf_counter() - t0
    busy = f.kernel_busy()
    ms, launches = f.kernel_times()
    share1 = f.share_stats()
    xchg_ms, xchg_n = f.exchange_time()		# hipEvents around the ncclGroup on the count/merge stream (native transport)
    exchange_ranks = sf.exchange_ranks() if sf is not None else 1	# ncclCommCount of the library's communicator
    # Frame mode, outside the timed region: what the exchange left on every rank.  No oracle here (the bench may not call it for the
    # measured path) -- two properties the exchange must have whatever the data: the reduced hit counts of the last frame hold every
    # spectrum of the frame exactly once per column (each rank's shard arrived, none twice), and the replicated state -- hit counts,
    # histogram, spectrum -- is bit-identical on all ranks.
    exchange_check = None
    if sf is not None:
        import hashlib
        sf.gather_state()
        if f.finish() < 0:
            raise RuntimeError("device error")
        hc = f.hitcount
        per_column = F * spb * world
        counts_ok = bool((hc.astype("int64").sum(0) == per_column).all())
        dig = hashlib.sha256(hc.tobytes() + f.histogram.tobytes() + f.spectrum.tobytes()).digest()[:8]
        mine = torch.tensor([int.from_bytes(dig, "little") >> 1, int(counts_ok)], dtype=torch.int64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
        if world > 1:
            dist.all_gather(every, mine)
        exchange_check = {"hit_counts_per_column": per_column,
                          "every_spectrum_counted_once_on_every_rank": all(int(e[1].item()) == 1 for e in every),
                          "replicated_state_bit_identical_across_ranks": all(int(e[0].item()) == int(every[0][0].item()) for e in every)}

    ms_all, n_all, iso, twin_ms = [0.0] * 3, [0] * 3, None, None
    if extra:
        # K2 / K3 durations in the pipeline (informational): a short extra pass with events around
        # every kernel, outside the timed region
        f.profile(1)
        run_steps(2)
        sync()
        f.kernel_times()
        run_steps(8)
        sync()
        ms_all, n_all = f.kernel_times()
        # K1 alone (same launches, nothing running beside it: one stream), reported as roofline.isolated
        if mode == "batch":
            f.set_overlap(False)
            run_steps(2)
            sync()
            f.kernel_times()
            run_steps(8)
            sync()
            ms_i, n_i = f.kernel_times()
            f.set_overlap(True)
            if n_i[0]:
                iso = ms_i[0] / n_i[0]
    f.profile(False)

    # The headline runs with relaxed input ordering (the caller promises to leave the samples alone until finish()); the same
    # steps with the default, strict ordering against the caller's stream are timed beside it, outside the timed region.
    strict_value = None
    if mode == "batch" and not args.strict_ordering and extra:
        f.set_input_ordering(True)
        run_steps(3)
        sync()
        strict_value = timed(max(4, min(steps, 40)))
        f.set_input_ordering(False)

    # the practical ceiling for K1 on this chip: its memory traffic (same loads, order, prefetch depth, stores)
    # without its arithmetic, measured live on the same buffers
    if mode == "batch" and name == "C2" and not args.no_traffic_twin and not light:
        try:
            sub = min(F, 64)
            twin_ms = f.traffic_twin(iq[:sub * samples_per_batch], sub, spb, reps=50)
        except Exception:
            twin_ms = None

    k1_busy_rank = busy[0] / max(1, launches[0])		# this rank's K1 union time per launch, ms
    k1_busy_ranks = [k1_busy_rank]
    xchg_ranks_all = [exchange_ranks]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gg = torch.zeros(world, 2, dtype=torch.float64, device="cuda")
        gg[rank, 0], gg[rank, 1] = k1_busy_rank, float(exchange_ranks)
        dist.all_reduce(gg, op=dist.ReduceOp.SUM)
        k1_busy_ranks = [float(v) for v in gg[:, 0].tolist()]
        xchg_ranks_all = [int(v) for v in gg[:, 1].tolist()]

    total_samples = world * steps * F * samples_per_batch
    value = total_samples / elapsed / 1e6

    out = None
    if rank == 0:
        n_k1 = max(1, launches[0])
        k1_ms = ms[0] / n_k1					# plain average of the individual launch durations
        k1_busy = busy[0] / n_k1				# union of the K1 intervals / launches
        samples_per_launch = steps * F * samples_per_batch / n_k1
        alg_bytes = bytes_per_sample * samples_per_launch
        achieved = alg_bytes / (k1_busy * 1e-3) / 1e9 if k1_busy > 0 else 0.0
        achieved_plain = alg_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        isolated = None
        if iso:
            a_i = alg_bytes / (iso * 1e-3) / 1e9
            isolated = {"k1_ms_per_launch": iso, "achieved": a_i, "frac": a_i / HBM_PEAK_GBS,
                        "note": "K1 with nothing running beside it (single stream), outside the timed region"}
        # HBM bytes per K1 launch from the PMC passes of the same command (tools/profile_round.sh writes the file):
        # only quoted when it was measured for this very launch shape
        traffic, traffic_src = None, None
        import glob
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k1_pmc*.json")), reverse=True):	# newest round first
            try:
                j = json.load(open(pmc))
            except Exception:
                continue
            for e in (j if isinstance(j, list) else [j]):
                if (e.get("config") == name and e.get("bins") == bins and
                        abs(e.get("samples_per_launch", 0) - samples_per_launch) < 1):
                    traffic, traffic_src = e.get("hbm_bytes_per_launch"), os.path.basename(pmc)
                    break
            if traffic:
                break
        k1_name = {10: "k1_fft_bin (K1, one wave per spectrum)",
                   13: "k1w_fft_bin (K1, 512 threads x 16 points per spectrum, radix 16.16.16.2 of FMA butterflies, overlap reused from registers)",
                   16: "k1h_fused (K1, radix-16 plan of FMA butterflies: two 256-point levels in one kernel, intermediate in the XCD's L2)"}[cfg["log2n"]]
        if cfg["log2n"] == 10 and os.environ.get("FOSPHOR_AMD_K1", "1")[:1] == "2":
            k1_name = "k1v2_fft_bin (K1, two waves per spectrum)"
        sub_b = samples_per_launch / samples_per_batch
        wf_rows = 1024
        roofline = {"bound": "hbm", "kernel": k1_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "k1_busy_ms_per_launch": k1_busy, "k1_ms_per_launch": k1_ms, "k1_launches": launches[0],
                    "k1_overlap": (ms[0] / busy[0]) if busy[0] > 0 else None,
                    "achieved_plain_average": achieved_plain, "frac_plain_average": achieved_plain / HBM_PEAK_GBS,
                    "accounting": "K1s of consecutive sub-launches run on two streams and overlap at their edges: achieved = "
                                  "algorithmic bytes of all K1 launches / time with at least one K1 running (union of the "
                                  "hipEvent intervals); *_plain_average divides by the mean individual duration, which counts "
                                  "the shared time twice",
                    "whole_path_frac": value * 1e6 * bytes_per_sample / 1e9 / HBM_PEAK_GBS / max(1, world),
                    # also against what the part delivers (6.29 TB/s float4 copy, MI355X_MICROARCH.md)
                    "frac_of_achievable": achieved / ACHIEVABLE_GBS,
                    "k1_traffic_rate_GBs": (traffic / 1e9 / (k1_busy * 1e-3)) if (traffic and k1_busy > 0) else None,
                    "k2_ms_per_launch": ms_all[1] / max(1, n_all[1]),
                    "k3_ms_per_launch": ms_all[2] / max(1, n_all[2]),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "unique_bytes_per_launch": alg_bytes / over,
                    "isolated": isolated,
                    "traffic_twin": None if not twin_ms else {
                        "ms_per_launch": twin_ms,
                        "k1_isolated_over_twin": (iso / twin_ms) if iso else None,
                        "note": "a kernel with K1's loads (same tile order, prefetch depth) and stores but no arithmetic, "
                                "same buffers: the practical floor the memory system sets for one K1 launch"}}
        if cfg["log2n"] == 13:
            # the form the FFT launches of the timed steps took (chosen at submit time from the state of the queue)
            roofline["fft_launch_form"] = {"shared": share1[0] - share0[0], "full_chip": share1[1] - share0[1], "shared_work_groups": share1[2]}
        workload = cfg["text"] % dict(bins=bins, bps=F, msamp=F * samples_per_batch >> 20,
                                      mib=F * samples_per_batch * bytes_per_sample >> 20, sub=int(round(sub_b)))
        if light:
            out = {"metric": "complex IQ MSamples/s @%d-pt FFT" % n_fft, "value": value, "unit": "MSamples/s",
                   "steps": steps, "warmup": warmup, "ms_per_step": elapsed * 1e3 / steps, "dtype": "f32",
                   "workload": workload, "batches_per_step": F, "precondition_s": precondition_s,
                   "input_ordering": "strict" if args.strict_ordering else "relaxed",
                   "host_submit_fraction": t_submit / elapsed,
                   "roofline": {k: roofline[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "k1_busy_ms_per_launch",
                                                         "k1_ms_per_launch", "k1_launches", "whole_path_frac", "algorithmic_bytes_per_launch",
                                                         "unique_bytes_per_launch") }}
            if "fft_launch_form" in roofline:
                out["roofline"]["fft_launch_form"] = roofline["fft_launch_form"]
        else:
            out = {
                "metric": "complex IQ MSamples/s @%d-pt FFT" % n_fft,
                "value": value, "unit": "MSamples/s",
                "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": elapsed * 1e3 / steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {
                    "workload": workload,
                    "mode": mode, "batches_per_step": F, "spectra_per_batch": spb, "ring_steps": ring,
                    "k1_launches_per_step": n_k1 / max(1, steps),
                    "input": "white complex Gaussian sigma=0.05, %s IQ resident in HBM (%d MiB ring, written once)"
                             % ("fp16" if cfg["fp16"] else "fp32", iq.numel() * iq.element_size() >> 20),
                    "precondition_s": precondition_s, "precondition_steps": pre_steps,
                    "placement": placement,	# allocations re-rolled before the run until the FFT kernel's memory twin ran in its fast state

                    "waterfall": "dead-store rule: a row that a later spectrum of the same call overwrites is not stored, "
                                 "so a step stores the rows of its last %d of %d spectra (the ring ends in the same state; "
                                 "the reference would store all of them)" % (min(wf_rows, F * spb), F * spb),
                    "input_ordering": "strict" if args.strict_ordering else "relaxed",
                    "strict_ordering_value": strict_value,	# MSamples/s of the same steps with the default (strict) ordering, untimed extra pass
                    "host_submit_fraction": t_submit / elapsed,
                    # multi-GPU self-description: what the exchange actually spanned, as the transport itself reports it
                    "transport": ("none" if (sf is None or not sf.active) else
                                  "native RCCL (library communicator)" if sf.exchange == "rccl" else "torch.distributed (%s)" % dist.get_backend()),
                    "transport_agreement": transport,
                    "exchange_ranks": exchange_ranks,			# ncclCommCount(library communicator) on rank 0 (1 = no exchange)
                    "exchange_ranks_per_rank": xchg_ranks_all,
                    "exchange_check": exchange_check,		# frame mode: properties of the reduced arrays, checked after the timed region
                    "exchange_ms_per_frame": (xchg_ms / xchg_n) if xchg_n else None,	# hipEvents around the ncclGroup, rank 0
                    "exchanges_timed": xchg_n,
                    "k1_busy_ms_per_launch_per_rank": {"min": min(k1_busy_ranks), "max": max(k1_busy_ranks), "all": k1_busy_ranks},
                    "exchange": "none" if (sf is None or not sf.active) else
                                ("native RCCL (%s) of hit counts / live sum / max once per frame of %d batches per GPU, on the "
                                 "library's count/merge stream" % ("reduce-scatter + sliced merge" if sf.sliced else
                                                                   "one ncclGroup of three all-reduces", F))
                                if sf.exchange == "rccl" else
                                "torch.distributed all-reduces (RCCL) of hit counts / live sum / max once per frame of %d batches per GPU" % F,
                },
                "roofline": roofline,
            }

    # orderly shutdown on every rank: the library's communicator and instance first (torch's process group: the caller)
    try:
        if sf is not None:
            sf.close()
        else:
            f.close()
    except Exception as e:
        sys.stderr.write("rank %d: shutdown: %s\n" % (rank, e))
    del iq
    torch.cuda.empty_cache()
    return out



PCIE_GEN5_X16_GBS = 63.0	# Gen5 x16, one direction, before protocol overhead


def measure_sink(ctx, seconds=2.0):
    """PCIe-INCLUSIVE rates of the drop-in path (SURVEY 8d's "second number with H2D included"; never `value`), C1/C2 geometry
    (1024-pt FFT, the reference's 128 bins), host samples in pageable memory as a GNU Radio buffer would be:
      work_fifo      gr::fosphor::base_sink_c::work() -> fifo -> render(): fosphor_process() (base_sink_c_impl.cc:130-201,432-462)
                     as the GNU-Radio-free runtime does it (fosphor_sink.cpp: pinned FIFO, uploads on their own stream, up to 8 batches
                     per call), fed by a native thread in work() calls of 64 Ki samples for >= `seconds` (and of 1 Mi samples for half that);
      process_calls  plain fosphor_process(self, samples, 1 Mi) calls from host memory (the literal reference call shape,
                     base_sink_c_impl.cc:146-175 -> cl.c:903-910), one fosphor_draw() per 8 calls, for >= `seconds`."""
    import ctypes as C
    import numpy as np
    pkg = ctx["pkg"]
    L = pkg.load()
    out = {"unit": "MSamples/s", "link": "PCIe Gen5 x16, %.0f GB/s one way" % PCIE_GEN5_X16_GBS,
           "note": "host-resident fp32 IQ, H2D inside the timed region; never the headline value"}
    n = 32 << 20						# 32 Mi samples = 256 MiB per pass
    x = (np.random.default_rng(3).standard_normal((n, 2)) * 0.05).astype(np.float32)
    # (a) work() -> FIFO -> upload -> kernels
    s = L.fosphor_amd_sink_new_len(1 << 24)
    try:
        if L.fosphor_amd_sink_start(s) != 1:
            raise RuntimeError("sink did not start")
        if L.fosphor_amd_sink_feed(s, x.ctypes.data, n, 64 * 1024, 1) < 0:	# warm-up (boot, staging buffers)
            raise RuntimeError("sink consumed nothing")
        def leg(chunk, budget):
            reps, dt = 0, 0.0
            while dt < budget:
                d = L.fosphor_amd_sink_feed(s, x.ctypes.data, n, chunk, 4)
                if d < 0:
                    raise RuntimeError("sink stalled")
                dt += d
                reps += 4
            v = n * reps / dt / 1e6
            return {"value": v, "seconds": dt, "samples": n * reps, "frac_of_link": v * 8e6 / (PCIE_GEN5_X16_GBS * 1e9),
                    "dropped": int(L.fosphor_amd_sink_dropped(s)), "work_call_samples": chunk, "fifo_samples": 1 << 24}
        out["work_fifo"] = leg(64 * 1024, seconds)
        # the same with work() handed 1 Mi samples at a time (one memcpy of 8 MiB per call: what bounds the single producer thread
        # at 64 Ki is its per-call cost, not the link)
        out["work_fifo_1Mi_calls"] = leg(1 << 20, seconds / 2)
        L.fosphor_amd_sink_stop(s)
    finally:
        L.fosphor_amd_sink_free(s)
    # (b) fosphor_process() per 1 Mi samples
    f = pkg.Fosphor()
    try:
        m = 1 << 20
        for k in range(8):
            assert f.process(x[k * m:(k + 1) * m]) == 0
        f.draw()
        t0 = time.perf_counter()
        calls = 0
        while True:
            for k in range(8):
                o = ((calls + k) % (n // m)) * m
                if f.process(x[o:o + m]) != 0:
                    raise RuntimeError("fosphor_process failed")
            f.draw()
            calls += 8
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
        v = calls * m / dt / 1e6
        out["process_calls"] = {"value": v, "seconds": dt, "samples": calls * m, "frac_of_link": v * 8e6 / (PCIE_GEN5_X16_GBS * 1e9),
                                "call_samples": m, "draw_every_calls": 8}
    finally:
        f.close()
    return out

def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher started us: become the launcher's parent (child process; nothing in THIS process has touched the GPU)
        sys.exit(self_launch(args))
    # The library's two FFT streams must not share a hardware queue with each other or with RCCL's streams
    # (HIP maps streams onto 4 queues by default; DESIGN_HISTORY.md section 5 "Hardware queues"): neutral at N=1 (measured).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FOSPHOR_BENCH_ONE_GPU"):		# test hook: every rank on device 0 (tests/test_gpu_dist.py, with the gloo backend)
        local_rank = 0
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d: measuring with %d ranks\n" % (args.gpus, world, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # (FOSPHOR_BENCH_BACKEND=gloo: test hook -- several ranks sharing ONE GPU, which RCCL refuses; tests/test_gpu_dist.py)
        backend = os.environ.get("FOSPHOR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from _pkg import gr_fosphor_amd
    ctx = {"torch": torch, "dist": dist, "pkg": gr_fosphor_amd, "world": world, "rank": rank}

    out = measure(args.config, args, ctx, args.steps, args.warmup)

    if world == 1 and args.config == "C2" and not args.no_other_configs and args.mode in ("auto", "batch") \
            and not args.bins and not args.batches_per_step:
        # The other single-GPU BASELINE configurations, in the same process, so that the driver's clock has seen them too
        # (each: its own instance and input ring, a short pre-conditioning, `--other-steps` timed steps of bench.py --config Cx)
        others = {}
        # (a C5 step is one 0.29 ms frame: ten times the steps, so that its timed region is tens of milliseconds like C3's)
        for name, mult in (("C3", 1), ("C5", 10)):
            try:
                others[name] = measure(name, args, ctx, args.other_steps * mult, 5 * mult, light=True)
            except Exception as e:
                others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            # C3 in calls of 4 batches (the call size of rounds 1-3; too few tiles for the FFT launch to share the chip)
            c3_4 = measure("C3", args, ctx, args.other_steps, 5, light=True, batches_per_step=4)
            if isinstance(others.get("C3"), dict) and "value" in others["C3"]:
                others["C3"]["four_batch_call_value"] = c3_4["value"]
        except Exception as e:
            sys.stderr.write("bench.py: C3 with 4-batch calls: %s\n" % e)
        try:
            others["sink"] = measure_sink(ctx, args.sink_seconds)
        except Exception as e:
            others["sink"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if out is not None:
            out["other_configs"] = others

    guard = None
    if rank == 0 and out is not None:
        # no silent fallback in a scaling record: anything but the native exchange over all N ranks fails the run (exit code 3)
        guard = scaling_record_guard(world, out["config"].get("transport"), out["config"].get("exchange_ranks_per_rank", []),
                                     test_hook=os.environ.get("FOSPHOR_BENCH_BACKEND", "nccl") != "nccl",
                                     check=out["config"].get("exchange_check"))
        if guard:
            out["invalid"] = guard
            sys.stderr.write("bench.py: %s\n" % guard)
    if rank == 0 and out is not None:
        if world == 1 and not args.no_cpu_baseline and args.config == "C2":
            out["cpu_baseline"] = cpu_baseline((args.bins or CONFIGS["C2"]["bins"]), args.cpu_seconds)
        # RCCL prints a version banner through C stdio, which would otherwise be flushed after this line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if guard:
        sys.exit(3)


if __name__ == "__main__":
    main()
