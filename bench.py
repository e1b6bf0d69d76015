#!/usr/bin/env python3
"""Headline benchmark: complex IQ MSamples/s through the fosphor hot path at 1024-pt FFT.

Contract (one JSON line on rank 0):
    python bench.py --gpus N --steps K --warmup W
    N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A STEP is one batch of 1024 spectra x 1024 points (1 Mi complex samples, 8 MiB of fp32 IQ) per
GPU, already resident in HBM, taken through the whole path: windowed FFT -> log-power -> exact
histogram bin -> hit counts -> persistence-histogram rise/decay, live EMA, max-hold, waterfall.
Workload = BASELINE.json configs[1] ("C2": 1024-pt FFT, batch=1024, 1024x256 histogram +
waterfall, 1xMI355X); per-GPU work is the same at every N (weak scaling, configs[3] "C4").

  N = 1  ("batch" mode): every step gets its own state update, exactly like successive
         fosphor_process() calls of the reference (cl.c:870-968); steps are submitted
         --batches-per-launch at a time (fosphor_amd_process_device), which changes launch
         granularity, not results.
  N > 1  ("frame" mode): the spectra of a display frame (--batches-per-launch steps per GPU, default
         256 = 1.1 ms of compute; a 60 Hz display frame would be 16 ms) are time-sharded over the
         ranks; hit counts / live sums / max are all-reduced over RCCL once per frame and every rank
         applies the same state update.  The all-reduce of frame k overlaps the FFT of
         frame k+1.  FOSPHOR_AMD_FORCE_EXCHANGE=1 runs the collectives on a single rank (smoke test
         of the RCCL path: stream ordering, library-owned buffers).

The input ring is larger than the 256 MiB Infinity Cache so IQ reads come from HBM.
The defaults (131072 steps = 137 G samples, ~0.3 s) are long enough to be past the first
milliseconds of a run, during which the clocks are still settling and K1 runs 10-30 % slower
(measured: 1280 steps 390 GS/s, 32768 steps 450-459, 131072 steps 462-464, 524288 steps 457-461).

roofline: the dominant kernel is K1 (fft_bin).  achieved = 8 B x samples per launch / mean K1
duration, measured with hipEvents on the library's stream inside the timed region.
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

N_FFT = 1024
BATCH = 1024
HBM_PEAK_GBS = 8000.0		# MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
BYTES_PER_SAMPLE = 8		# algorithmic read, one complex fp32 sample


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=131072)
    ap.add_argument("--warmup", type=int, default=4096)
    ap.add_argument("--bins", type=int, default=256)
    ap.add_argument("--batches-per-launch", type=int, default=0,
                    help="steps per launch (batch mode, default 64) / per display frame and GPU (frame mode, default 256)")
    ap.add_argument("--ring-batches", type=int, default=0, help="distinct batches of IQ resident in HBM (8 MiB each); default 2 launches")
    ap.add_argument("--mode", choices=["auto", "batch", "frame"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic-twin", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(bins, seconds):
    """Oracle (CPU restatement of the reference kernels) on all host cores, bounded sample."""
    import numpy as np
    from oracle_lib import Oracle, build_oracle, gaussian_iq
    build_oracle(ref=False)
    cores = os.cpu_count() or 1
    nthreads = min(cores, 64)			# the display stage has 64 column groups (cl.c:945-950)
    o = Oracle(n_bins=bins)
    x = gaussian_iq(BATCH * N_FFT, 7)
    o.process(x, nthreads=nthreads)		# warm-up, page in
    n = 0
    t0 = time.perf_counter()
    while True:
        o.process(x, nthreads=nthreads)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 4096:
            break
    return {"value": n * BATCH * N_FFT / el / 1e6, "unit": "MSamples/s", "cores": nthreads, "kind": "port",
            "sample": "%d batches of %d x %d-pt spectra (%.1f s), oracle C restatement of fft.cl+display.cl, "
                      "%d threads of %d host cores" % (n, BATCH, N_FFT, el, nthreads, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("FOSPHOR_AMD_FORCE_EXCHANGE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from _pkg import gr_fosphor_amd
    from gr_fosphor_amd.dist import ShardedFosphor

    mode = args.mode if args.mode != "auto" else ("batch" if world == 1 else "frame")
    # batch mode: 64 steps per launch (more pushes the intermediates out of the Infinity Cache).
    # frame mode: a display frame of 256 steps per GPU (1.1 ms of compute; 60 Hz would be 16 ms): the state
    # update runs once per frame whatever its length, so a longer frame only makes the exchange and the host's
    # per-frame work (three collectives) rarer.
    F = args.batches_per_launch if args.batches_per_launch > 0 else (64 if mode == "batch" else 256)
    ring = args.ring_batches if args.ring_batches > 0 else 2 * F
    ring = max(F, (ring // F) * F)

    # synthetic white complex Gaussian IQ, sigma 0.05 per component, resident in HBM
    g = torch.Generator(device="cuda")
    g.manual_seed(7 + rank)
    iq = torch.empty((ring * BATCH * N_FFT, 2), dtype=torch.float32, device="cuda")
    iq.normal_(0.0, 0.05, generator=g)
    samples_per_batch = BATCH * N_FFT

    stream = torch.cuda.current_stream().cuda_stream
    if mode == "batch":
        f = gr_fosphor_amd.Fosphor(n_bins=args.bins, max_spectra=F * BATCH, max_batches=F, stream=stream)
        sf = None
    else:
        sf = ShardedFosphor(gr_fosphor_amd.Fosphor, rank, world, n_bins=args.bins, max_spectra=F * BATCH)
        f = sf.f

    def run_steps(n_steps, pos):
        """submit n_steps batches starting at ring position pos; returns new pos"""
        done = 0
        while done < n_steps:
            nb = min(F, n_steps - done)
            if pos + nb > ring:
                pos = 0
            view = iq[pos * samples_per_batch:(pos + nb) * samples_per_batch]
            if mode == "batch":
                rv = f.process_device(view, nb, BATCH)
                if rv:
                    raise RuntimeError("process_device -> %d" % rv)
            else:
                sf.frame(view, nb * BATCH * world, overlap=True, wait_producer=False)	# IQ was generated before the warm-up
            pos += nb
            done += nb
        if sf is not None:
            sf.flush()
        return pos

    def sync():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize()

    f.finish()				# instance boot (table uploads, initial fills: cl.c:981-995) is not a step
    pos = run_steps(args.warmup, 0)
    sync()
    # timed region: hipEvents around K1 only (events around K2/K3 too cost ~6 % of throughput:
    # they sit on the critical path of the count/merge streams)
    if not os.environ.get("BENCH_NO_PROFILE"):	# debugging aid: cost of the hipEvents themselves
        f.profile(2)
    t0 = time.perf_counter()
    run_steps(args.steps, pos)
    t_submit = time.perf_counter() - t0		# host time to queue everything (host-bound if ~ elapsed)
    sync()
    elapsed = time.perf_counter() - t0
    ms, launches = f.kernel_times()

    # K2 / K3 durations in the pipeline (informational): a short extra pass with events around
    # every kernel, outside the timed region
    f.profile(1)
    run_steps(8 * F, 0)
    f.kernel_times()
    run_steps(32 * F, 0)
    ms_all, n_all = f.kernel_times()

    # K1 alone (same launches, K2/K3 not running beside it): a short extra pass outside the
    # timed region, reported as roofline.isolated
    iso = None
    if mode == "batch":
        f.set_overlap(False)
        run_steps(8 * F, 0)
        f.kernel_times()
        run_steps(32 * F, 0)
        ms_i, n_i = f.kernel_times()
        f.set_overlap(True)
        if n_i[0]:
            iso = ms_i[0] / n_i[0]
    f.profile(False)

    # the practical ceiling for K1 on this chip: its memory traffic (same loads, order, prefetch depth, stores)
    # without its arithmetic, measured live on the same buffers
    twin_ms = None
    if mode == "batch" and not args.no_traffic_twin:
        try:
            twin_ms = f.traffic_twin(iq[:F * samples_per_batch], F, BATCH, reps=50)
        except Exception:
            twin_ms = None

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_samples = world * args.steps * samples_per_batch
    value = total_samples / elapsed / 1e6

    if rank == 0:
        k1_ms = ms[0] / max(1, launches[0])
        samples_per_launch = args.steps * samples_per_batch / max(1, launches[0])
        achieved = BYTES_PER_SAMPLE * samples_per_launch / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_k1_pmc.json")
        isolated = None
        if iso:
            a_i = BYTES_PER_SAMPLE * F * samples_per_batch / (iso * 1e-3) / 1e9
            isolated = {"k1_ms_per_launch": iso, "achieved": a_i, "frac": a_i / HBM_PEAK_GBS,
                        "note": "K1 with K2/K3 not running beside it (single stream), outside the timed region"}
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("batches_per_launch") == F and j.get("bins") == args.bins:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        k1_name = "k1v2_fft_bin (K1, two waves per spectrum)" if os.environ.get("FOSPHOR_AMD_K1", "1")[:1] == "2" \
            else "k1_fft_bin (K1, one wave per spectrum)"
        out = {
            "metric": "complex IQ MSamples/s @1024-pt FFT",
            "value": value, "unit": "MSamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "C2: 1024-pt FFT, batch=1024 spectra/step/GPU, 1024x%d histogram + waterfall" % args.bins,
                "mode": mode, "batches_per_launch": F, "ring_batches": ring,
                "input": "white complex Gaussian sigma=0.05, fp32 IQ resident in HBM (%d MiB ring)" % (ring * 8),
                "host_submit_fraction": t_submit / elapsed,
                "exchange": "none" if world == 1 else "RCCL all-reduce of hit counts / live sum / max once per frame of %d steps" % F,
            },
            "roofline": {"bound": "hbm", "kernel": k1_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "k1_ms_per_launch": k1_ms, "k1_launches": launches[0],
                         "k2_ms_per_launch": ms_all[1] / max(1, n_all[1]),
                         "k3_ms_per_launch": ms_all[2] / max(1, n_all[2]),
                         "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * samples_per_launch,
                         "isolated": isolated,
                         "traffic_twin": None if not twin_ms else {
                             "ms_per_launch": twin_ms,
                             "k1_isolated_over_twin": (iso / twin_ms) if iso else None,
                             "note": "a kernel with K1's loads (same tile order, prefetch depth) and stores but no arithmetic, "
                                     "same buffers: the practical floor the memory system sets for one K1 launch"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.bins, args.cpu_seconds)
        # RCCL prints a version banner through C stdio, which would otherwise be flushed after this line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)

    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
