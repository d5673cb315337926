#!/usr/bin/env python3
"""Benchmark of the jamming-detection DSP path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic antenna capture per GPU
(BASELINE.json configs[1]; one stream per rank = configs[4] for N > 1):
  one fused HBM pass for K1 per-chunk power (+ 5th-percentile/+6 dB threshold), K3 amplitude
  statistics and K4 onset; K2 fused unpack + 4096-pt Welch PSD (1-s chunks); the rank's
  onset-aligned 2^19-sample slice cut into a TDOA slot; ONE RCCL all-gather of the slots; EVERY
  antenna pair solved (2^20-pt FFT cross-correlations), the pairs dealt over the ranks so that
  each rank runs one multi-pair K5 launch of constant size; an RCCL gather of the per-stream
  result vectors (with the solved pairs inside) to rank 0.
  At N = 1 the K5 solve is BASELINE configs[3]: this capture's slot against the slots of two
  further antennas (prepared before the timed region, as if gathered) = 3 antennas, 3 pairs.
  K2 (VALU/LDS bound) runs on one HIP stream, the HBM-bound scan, the gathers and K5
  concurrently on a second one; they join before the result vector is packed.
Captures are generated in HBM before the timed region (2^30 bytes = 536 870 912 I/Q samples
per GPU, integer-only generator, seeds 1234 + rank) -- inputs are resident when timing starts.

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (K2 welch_kernel<4096>
+ its finalize): algorithmic bytes = 2 B x samples per launch, duration from HIP events
recorded on the launch stream around every K2 launch of the timed steps; ``roofline_valu``
prices the same launch against the measured VALU issue ceiling (what actually bounds it);
``secondary`` holds the scan and the K5 solve.  ``self_check`` validates the gathered results
against what the synthetic captures were built to contain (known delays, known burst span), so a
wrong multi-GPU exchange cannot print a plausible number.  ``cpu_baseline`` times the
numpy/scipy oracle (oracle/gpsjam_oracle.py, kind "port") on a bounded prefix of the same
capture on the host's cores (rank 0 at N = 1; with --cpu-baseline-all-ranks one process per rank at N > 1).  ``end_to_end`` (N = 1) is the
file / host buffer -> result rate with PCIe included; it is never ``value``.
"""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "gps-jamming_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured copy rate
CAPTURE_BYTES = 1 << 30
NPERSEG = 4096
CHUNK_SAMPLES = 2048000
SLICE = 1 << 19
DELAYS = (0, 3, -5, 7, -2, 4, -6, 1)
# per-antenna burst gain: every antenna must clear K4's 50x-noise onset rule with margin
# (sigma_jam > 7 sigma_noise = 43.75 LSB), so the gains stay near 1 (sigma 60 / 54 / 51 / 57 LSB)
JAM_GAIN = (1.0, 0.9, 0.85, 0.95)
JAM_SPAN = (0.4, 0.7)      # burst in source time, as fractions of the capture
# the reference's own operating point (VERDICT r03 missing 3): FFT_SIZE = 1024 (skrypty/widmo_plot.py:10,48), TDOA
# slices of 50 000 samples (skrypty/triangulateTDOA.py:26), three antenna captures of 10 s (worker.py:184-196, 586-600)
REF_NPERSEG = 1024
REF_SLICE = 50000
REF_CAPTURE_BYTES = 40960000


_LINE_FD = None


def guard_stdout():
    """Rank 0 prints ONE JSON line on stdout and nothing else may: RCCL, for one, writes a version banner to stdout
    when its first communicator is made.  From here on file descriptor 1 of this process points at stderr (so does
    everything a library prints), and `emit` writes the line to the descriptor stdout had."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, data)


def stream_spec(StreamSpec, antenna, nsamp):
    return StreamSpec(seed=1234, antenna=antenna, delay=DELAYS[antenna % len(DELAYS)],
                      jam_start=int(JAM_SPAN[0] * nsamp), jam_end=int(JAM_SPAN[1] * nsamp), noise_sigma=6.25,
                      jam_sigma=60.0 * JAM_GAIN[antenna % 4])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--capture-bytes", type=int, default=CAPTURE_BYTES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-reference-point", action="store_true",
                    help="skip the secondary figures at the reference's own operating point (nperseg 1024, 3 x 10-s deployment)")
    ap.add_argument("--cpu-baseline-all-ranks", action="store_true",
                    help="N > 1: every rank times the oracle on a prefix of its capture (default: the CPU baseline is N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--transport", default="torch", choices=("torch", "rccl"),
                    help="collectives through torch.distributed, or through the library's own gj_comm_* (RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo) on a one-GPU box")
    ap.add_argument("--precondition", type=int, default=30,
                    help="untimed steps run before the warm-up to settle clocks (not counted in --warmup)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the scan/TDOA kernels on the K2 stream instead of concurrently on a second one")
    ap.add_argument("--cpu-sample-chunks", type=int, default=None,
                    help="1-s chunks of the capture given to the CPU oracle (default 24 at N = 1, 8 per rank otherwise)")
    ap.add_argument("--split", action="store_true",
                    help="strong scaling (SURVEY 8(e)): --antennas captures of --capture-bytes each are cut into parts "
                         "over the N GPUs (gpsjam.split) instead of one capture per GPU")
    ap.add_argument("--emulate-rank", type=int, default=0,
                    help="--split --emulate-world W: which rank of the W this GPU plays (rank 0 also gathers and combines)")
    ap.add_argument("--antennas", type=int, default=3,
                    help="captures in --split mode (the reference's deployment has three, worker.py:586-600)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: form a process group of ONE (--backend, default nccl = RCCL) and issue the slot "
                         "all-gather and the result gather of the N > 1 path anyway -- the collective call path on one GPU")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="N = 1 only, rehearsal: give this GPU the per-rank load of a W-antenna deployment -- W - 1 further "
                         "slots prepared before the timed region, rank 0's share of the pairs (W / 2 of them) instead of "
                         "all; with --force-exchange the collectives are issued too.  Not the reported configuration")
    ap.add_argument("--pack-on-side", action="store_true",
                    help="one capture per GPU: pack the result vector on the second stream (measured neutral on the 1-GiB step: NOTES_r05)")
    ap.add_argument("--pack-on-main", action="store_true",
                    help="--split: pack the part vectors on the main stream behind K2 (round 4's order) instead of on the second stream")
    ap.add_argument("--side-priority", type=int, default=0, help="HIP priority of the second stream (0 default, -1 high)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="diagnostic: ranks only form the process group, all-reduce one number and print it")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds the self-launched ranks of --gpus N > 1 may take before they are stopped")
    ap.add_argument("--no-diagnosis", action="store_true",
                    help="self-launched ranks that fail are NOT followed by one fresh --rendezvous-only run")
    ap.add_argument("--fail-rank", type=int, default=-1,
                    help="diagnostic (tests): this rank exits with status 3 once the process group has formed")
    ap.add_argument("--no-probe", action="store_true",
                    help="N > 1 over nccl: make RCCL the default group at once instead of probing it first in a child process per rank")
    ap.add_argument("--mixed-groups", action="store_true",
                    help="rehearsal, with --force-exchange at N = 1: form the groups as an N > 1 run does (gloo control group + an "
                         "RCCL group of its own for the exchange) although there is one rank and nothing to probe")
    ap.add_argument("--inject-probe-failure", action="store_true",
                    help="diagnostic (tests): the RCCL probe children exit with status 3, as if RCCL could not form a group")
    ap.add_argument("--probe-fail", action="store_true", help=argparse.SUPPRESS)       # what --inject-probe-failure hands the child
    ap.add_argument("--probe-bytes", type=int, default=0,
                    help="--rendezvous-only: also all-gather this many bytes per rank over the group (the slot exchange's size)")
    ap.add_argument("--probe-timeout", type=float, default=150.0, help="seconds one RCCL probe child may take")
    ap.add_argument("--groups-only", action="store_true",
                    help="diagnostic: form the process groups exactly as a run would (control group, RCCL probe, data group or "
                         "fallback), all-gather one small row over the data group, print what carried it, and stop -- runs without a GPU")
    ap.add_argument("--fallback-of", default=None,
                    help="set by the launcher when this run replaces one whose RCCL group could not form: goes into the line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process never touches the GPU, it starts one rank per GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout, args))

    guard_stdout()
    # (The step's streams -- two, three on rank 0 of --split -- must not share a hardware queue: the runtime deals streams
    # over GPU_MAX_HW_QUEUES queues, four by default, and two on one queue run one after the other.  The pipelines test
    # for that when they make their streams: gpsjam/streams.py.)
    import numpy as np
    import torch
    import gpsjam
    from gpsjam.sharded import AntennaStream
    from gpsjam.synth import StreamSpec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local_rank = 0
    if args.rendezvous_only:
        raise SystemExit(rendezvous_only(args, torch, world, rank, local_rank))
    grouped = world > 1 or args.force_exchange     # a process group exists (of one, with --force-exchange)
    if args.force_exchange and world == 1 and "MASTER_ADDR" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
    # the groups are formed BEFORE this process touches the GPU: the RCCL probe runs in child processes (form_groups)
    G = form_groups(args, torch, world, rank, local_rank) if grouped else Groups()
    dist = G.dist
    if args.groups_only:
        raise SystemExit(groups_only(args, torch, G, world, rank))
    torch.cuda.set_device(local_rank)

    if grouped and args.fail_rank == rank:
        print(f"[bench] rank {rank}: --fail-rank, leaving with status 3", file=sys.stderr, flush=True)
        os._exit(3)

    status = 0
    dev = gpsjam.Device(local_rank)
    # one explicit HIP stream for everything in the step: the gpsjam kernels, torch's small
    # packing ops and the events that time the dominant kernel
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    dev.set_stream(work_stream.cuda_stream)
    if args.split:
        run_split(args, np, torch, gpsjam, G, dev, work_stream, world, rank)
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        dev.close()
        return
    nbytes = args.capture_bytes
    nsamp = nbytes // 2
    cap = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dev.synth_dev(stream_spec(StreamSpec, rank, nsamp), nsamp, cap)

    # N = 1: the slots of antennas 1 and 2 (BASELINE configs[3]), cut by the same kernels from their
    # own captures before the timed region -- on N ranks they would arrive through the slot gather
    aux, aux_onsets = None, []
    n_aux = max(args.emulate_world - 1, 2) if args.emulate_world else 2
    only_pairs = None
    if args.emulate_world:
        from gpsjam.sharded import pairs_of_rank
        only_pairs = pairs_of_rank(0, args.emulate_world)
    if world == 1:
        sb = dev.tdoa_slot_bytes(SLICE)
        aux = torch.zeros((n_aux, sb), dtype=torch.uint8, device="cuda")
        tmp = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        d_on = torch.zeros(4, dtype=torch.int64, device="cuda")
        for a in range(1, n_aux + 1):
            dev.synth_dev(stream_spec(StreamSpec, a, nsamp), nsamp, tmp)
            dev.onset_dev(tmp, nbytes, 200000, 1000, 50.0, d_on)
            dev.tdoa_slot_dev(tmp, nbytes, d_on, SLICE, aux[a - 1])
            torch.cuda.synchronize()
            aux_onsets.append(int(d_on[0].item()))
        del tmp
    stream = AntennaStream(dev, cap, nperseg=NPERSEG, chunk_samples=CHUNK_SAMPLES, slice_samples=SLICE,
                           rank=rank, world_size=world, overlap=not args.no_overlap, aux_slots=aux,
                           transport=args.transport, exchange_always=args.force_exchange and world == 1,
                           pairs=only_pairs if world == 1 else None, side_priority=args.side_priority,
                           pack_on_side=args.pack_on_side, group=G.data)
    torch.cuda.synchronize()

    def barrier():
        if grouped:
            dist.barrier()

    # clock / power-state conditioning, not part of the W warm-up steps: the first launches after
    # an idle period run at transient clocks (K2 varies 1.3 -> 1.8 -> 1.4 ms over the first ~20)
    for _ in range(args.precondition):
        stream.step()
    for _ in range(args.warmup):
        stream.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # N > 1: events on the SECOND stream around the exchange chain of every step (slot, slot all-gather, K5, pack, result
    # gather) -- what the first multi-GPU curve needs to be read rank by rank (per_rank below)
    ex = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps if grouped else 0)]
    t0 = time.perf_counter()
    gathered = None
    for k in range(args.steps):
        # the step, with HIP events on the launch stream around the dominant kernel
        stream.stream_scan()            # K1 + K3 + K4 in one HBM pass, + noise-floor threshold (second stream)
        ev[k][0].record()
        stream.welch()                  # K2
        ev[k][1].record()
        if ex:
            ex[k][0].record(stream._side)
        stream.tdoa()                   # slot, slot all-gather, this rank's share of the pairs (second stream)
        gathered = stream.exchange(0)   # pack (after the join) + result gather, issued on the second stream
        if ex:
            ex[k][1].record(stream._side)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0     # this rank's own steps, before it waits for the others
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # solo figures (nothing else in flight), after the timed region: K2 next to its as-run figure,
    # which includes whatever the concurrently running scan/TDOA kernels cost it; the scan and the K5 solve
    def timed(fn, on, reps=5):
        out = []
        for _ in range(3):              # clocks and caches as in a steady run
            fn()
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record(on)
            fn()
            b.record(on)
            torch.cuda.synchronize()
            out.append(a.elapsed_time(b))
        return sum(out) / len(out)

    side = stream._side
    solo_ms = timed(stream.welch, work_stream)
    scan_ms = timed(stream.stream_scan, side)
    d_on1 = torch.zeros(4, dtype=torch.int64, device="cuda")
    onset_alone_ms = timed(lambda: stream.dev_side.onset_dev(cap, nbytes, 200000, 1000, 50.0, d_on1), side)
    k5_ms = None
    if rank == 0 and stream.pairs:
        scratch = [torch.empty_like(t) for t in (stream.lags, stream.peaks, stream.margins)]

        def k5():                       # the same launch as in the step, into scratch outputs
            stream.dev_side.xcorr_slots_dev(stream.slots, stream.slot_bytes, stream.n_ant, SLICE, stream.pairs, *scratch)
        k5_ms = timed(k5, side)

    # one cold acquisition search (32 PRNs x 71 Doppler bins x 10 ms) on the quiet start of the
    # capture -- nothing acquires there, so every PRN runs all ten integration steps (the worst case)
    acq_ms = None
    if rank == 0:
        try:
            from gpsjam import gnss
            srch = gnss.AcqSearch(dev)
            acq_ms = timed(lambda: srch.search_dev(cap, nbytes, 0), work_stream, reps=5)
            acq_found = sum(r.acquired for r in srch.results())
            acq_shape = (len(srch.prns), len(srch.freqs), srch.intg, srch.nsamp)
            srch.close()
        except Exception as e:          # a secondary figure must never cost the primary line
            print(f"[bench] acquisition search skipped: {e!r}", file=sys.stderr)
            acq_ms = None

    # the reference's own operating point, in the driver-run line: K2 at nperseg 1024 on the same capture, and the
    # three-antenna 10-s deployment (N = 1 only; secondary figures must never cost the primary line)
    ref_point = None
    # Not beside a live RCCL process group (--force-exchange): the deployment step is captured into a HIP graph, its side
    # streams come from torch's stream pool -- and so does the process group's internal stream.  When the two coincide,
    # the group's watchdog thread, polling the end event of a collective it has not yet seen complete (it looks every
    # 100 ms), asks about an event whose stream is being captured: hipErrorCapturedEvent, and torch ends the process
    # (seen once in ~15 runs of the short --force-exchange rehearsal, round 6).  gpsjam/local.py says the same to its users.
    if rank == 0 and world == 1 and not grouped and not args.no_reference_point:
        try:
            psd_ref = torch.empty((dev.welch_rows(nbytes, CHUNK_SAMPLES, REF_NPERSEG), REF_NPERSEG), dtype=torch.float32, device="cuda")
            dev.reserve(max(dev.welch_workspace(nbytes, CHUNK_SAMPLES, REF_NPERSEG), dev.welch_workspace(nbytes, CHUNK_SAMPLES, NPERSEG)))
            k2_ref_ms = timed(lambda: dev.welch_dev(cap, nbytes, CHUNK_SAMPLES, REF_NPERSEG, 2.048e6, psd_ref), work_stream)
            # K2 at the two sizes, INTERLEAVED (A/B/A/B, same stream, same repetitions, back to back), transform kernel
            # and finalize launch timed apart (gj_welch_timed_dev): BENCH_r04 had 1024 slower than 4096 because the two
            # were timed minutes apart with different things in between
            ab = {NPERSEG: [], REF_NPERSEG: []}
            for n_ab in (NPERSEG, REF_NPERSEG):
                dev.welch_timed_dev(cap, nbytes, CHUNK_SAMPLES, n_ab, 2.048e6, stream.psd if n_ab == NPERSEG else psd_ref)
            for _ in range(6):
                for n_ab in (NPERSEG, REF_NPERSEG):
                    ab[n_ab].append(dev.welch_timed_dev(cap, nbytes, CHUNK_SAMPLES, n_ab, 2.048e6,
                                                        stream.psd if n_ab == NPERSEG else psd_ref))
            ref_point = {"k2_ms": k2_ref_ms, "psd": psd_ref, "ab": ab, "deployment": deployment(np, torch, gpsjam, dev, StreamSpec)}
        except Exception as e:
            print(f"[bench] reference operating point skipped: {e!r}", file=sys.stderr)
            ref_point = None

    elapsed = G.reduce(torch, elapsed, "max")
    welch_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    per_rank = None
    if grouped:
        exchange_ms = sum(a.elapsed_time(b) for a, b in ex) / max(len(ex), 1)
        per_rank = G.per_rank(torch, world, {"k2_ms": welch_ms, "scan_ms": scan_ms, "exchange_ms": exchange_ms,
                                             "step_ms": local_elapsed / max(args.steps, 1) * 1e3})
    proof = exchange_proof(args, torch, G, dev, stream.comm, world, rank)

    # CPU baseline: every rank times the oracle on a prefix of ITS capture at the same moment
    cpu = None
    # the contract asks for the CPU baseline on rank 0 at N = 1 only; --cpu-baseline-all-ranks times one oracle process
    # per rank at N > 1 as well
    if not args.no_cpu_baseline and (world == 1 or args.cpu_baseline_all_ranks or args.cpu_sample_chunks):
        chunks = args.cpu_sample_chunks or (24 if world == 1 else 8)
        barrier()
        cpu = cpu_baseline(np, cap, chunks, stream, gathered if rank == 0 else None, world, ref_point if rank == 0 else None)
        if world > 1:
            cpu["value"] = G.reduce(torch, cpu["value"], "sum")
            cpu["cores"] = world
            cpu["sample"] = f"{world} processes at once, one per capture, each: " + cpu["sample"]

    if rank == 0:
        results, tdoa = gathered.unpack()
        total_samples = float(nsamp) * world * args.steps
        value = total_samples / elapsed / 1e6
        achieved = (nbytes / 1e9) / (welch_ms / 1e3) if welch_ms > 0 else 0.0
        pmc = pmc_summary(nbytes)
        onsets = [r.onset for r in results] + aux_onsets
        line = {
            "metric": "Msamples/s uint8 I/Q through PSD+TDOA xcorr",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "precondition_steps": args.precondition,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: fused uint8->complex64 + 4096-pt Welch PSD + jamming power "
                                   "threshold on 1 GiB synthetic I/Q per GPU (+ K3 amp stats, K4 onset, TDOA slot; "
                                   + ("configs[3]: 3 antennas / 3 pairs 2^20-pt xcorr solve on this GPU)" if world == 1
                                      else f"configs[4]: slots all-gathered over RCCL, all {len(tdoa.pairs)} pairs of "
                                           f"{world} antennas solved, dealt over the ranks; result vectors gathered to rank 0)"),
                       "capture_bytes_per_gpu": nbytes, "nperseg": NPERSEG, "chunk_samples": CHUNK_SAMPLES,
                       "xcorr_slice": SLICE, "xcorr_antennas": stream.n_ant, "xcorr_pairs": len(tdoa.pairs),
                       "streams": world, "sharding": "one capture per GPU", "backend": G.backend if world > 1 else None,
                       "transport": (G.label if args.transport == "torch" else args.transport) if world > 1 else None},
            **proof,
            **G.line_fields(),
            "forced_exchange": bool(args.force_exchange and world == 1),
            "emulated_world": int(args.emulate_world) if world == 1 else 0, "side_priority": args.side_priority,
            "roofline": {"bound": "hbm", "kernel": "welch_kernel<4096> + welch_finalize_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic"], "traffic_unit": "HBM bytes per launch",
                         "traffic_source": pmc["source"], "traffic_measured_on_this_source": pmc["matches_build"],
                         "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": welch_ms,
                         # as-run K2 scatters from launch to launch (what runs beside it, clocks): the spread of THIS sample
                         "launch_ms_spread": (lambda v: {"min": v[0], "median": v[len(v) // 2], "max": v[-1], "launches": len(v)})(
                             sorted(a.elapsed_time(b) for a, b in ev)) if ev else None,
                         "overlap": bool(stream.overlap),
                         "solo": {"avg_launch_ms": solo_ms, "achieved": (nbytes / 1e9) / (solo_ms / 1e3),
                                  "frac": (nbytes / 1e9) / (solo_ms / 1e3) / HBM_PEAK_GBS},
                         "note": "K2 is FP32-VALU/LDS bound, not HBM bound (DESIGN.md section 5); "
                                 "'achieved' is K2 as run in the timed steps, i.e. with the HBM-bound "
                                 "scan + TDOA kernels executing concurrently on a second stream when "
                                 "overlap is true; 'solo' is K2 with nothing else in flight"},
            "roofline_valu": valu_roofline(pmc, welch_ms, solo_ms),
            "roofline_flops": flops_roofline(nbytes, welch_ms, solo_ms),
            "secondary": {
                "stream_scan_kernel (K1+K3+K4 fused) + threshold + tail kernels, solo": {
                    "bound": "hbm", "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": scan_ms,
                    "achieved": (nbytes / 1e9) / (scan_ms / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (nbytes / 1e9) / (scan_ms / 1e3) / HBM_PEAK_GBS,
                    **family_traffic("profiles/r06_pmc_scan/summary.json", nbytes)},
            },
            "results": {"pairs": [list(p) for p in tdoa.pairs], "lags": tdoa.lags,
                        "lag_margins": [round(m, 4) for m in tdoa.margins], "onsets": onsets,
                        "jamming_ranges_rank0": results[0].jamming_byte_ranges()[:4],
                        "baseline_rank0": results[0].baseline, "amp_mean": [r.amp_mean for r in results]},
            "self_check": self_check(results, tdoa, onsets, nsamp, stream.n_ant, len(stream.pairs) if only_pairs else None,
                                     proof=proof, world=world, share_gpu=args.share_gpu),
            "host": host_info(),
        }
        line["secondary"]["gj_onset_dev alone (K4 through the fused pass + tail since round 6; its own kernel chain took 0.62 ms), solo"] = {
            "bound": "hbm", "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": onset_alone_ms,
            "achieved": (nbytes / 1e9) / (onset_alone_ms / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (nbytes / 1e9) / (onset_alone_ms / 1e3) / HBM_PEAK_GBS,
            "onset_equals_the_step's": bool(int(d_on1[0].item()) == results[0].onset)}
        if k5_ms is not None:
            # ingest 2N B per antenna + 32 L B per transform (four-step floor), A + P transforms
            L, P = 2 * SLICE, len(stream.pairs)
            A = len({a for p in stream.pairs for a in p})
            k5_bytes = 2 * SLICE * A + 32 * L * (A + P)
            line["secondary"][f"K5 xcorr solve on rank 0, {A} antennas / {P} pairs, L = 2^20, solo"] = {
                "bound": "hbm", "algorithmic_bytes_per_launch": k5_bytes, "avg_launch_ms": k5_ms,
                "achieved": (k5_bytes / 1e9) / (k5_ms / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (k5_bytes / 1e9) / (k5_ms / 1e3) / HBM_PEAK_GBS,
                **family_traffic("profiles/r06_pmc_xcorr/summary.json", nbytes),
                "note": "PMC: FETCH_SIZE / WRITE_SIZE count the L2's requests to the fabric, Infinity-Cache hits included "
                        "(MI355X_MICROARCH.md): 49 MB written = the two sets of spectra once each, 55 MB fetched = what the two "
                        "consuming launches need -- the floor of a three-launch four-step transform, not HBM traffic"}
        if acq_ms is not None:
            n_prn, n_freq, intg, acq_nsamp = acq_shape
            n_fft = intg * n_freq * (1 + n_prn) + n_prn
            line["secondary"]["acquisition search (SURVEY 8(f)-4), solo"] = {
                "what": f"{n_prn} PRNs x {n_freq} Doppler bins x {intg} ms non-coherent, {2 * acq_nsamp}-pt FFT . conj(code) . IFFT, "
                        f"peak test per step; quiet input: all {intg} steps run ({acq_found} false acquisitions)",
                "avg_search_ms": acq_ms, "transforms_per_search": n_fft,
                "transforms_per_s": n_fft / (acq_ms / 1e3),
                "real_time_factor": (intg * 1e-3) / (acq_ms / 1e3),
                "k2_transforms_per_s_for_scale": (nbytes / 2 / (NPERSEG // 2)) / (solo_ms / 1e3),
                **family_traffic("profiles/r04_pmc_acq/summary.json", nbytes),
                "parity": "unpinned (gnssdec unbuildable here); oracle = numpy restatement of sdracq.c / sdrcmn.c"}
        if ref_point is not None:
            par = (cpu or {}).pop("reference_point_parity", None)
            # the same measurement as its 4096 sibling in the interleaved block below (kernel + finalize, averaged over the
            # rounds): the two figures of one line come from the same minute of the same box
            k2r = (sum(k for k, _ in ref_point["ab"][REF_NPERSEG]) + sum(f for _, f in ref_point["ab"][REF_NPERSEG])) / len(ref_point["ab"][REF_NPERSEG])
            line["secondary"][f"K2 welch_kernel<{REF_NPERSEG}> + finalize at the reference's FFT size (widmo_plot.py:10,48), same capture, solo"] = {
                "bound": "hbm", "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": k2r,
                "achieved": (nbytes / 1e9) / (k2r / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (nbytes / 1e9) / (k2r / 1e3) / HBM_PEAK_GBS, "msamples_per_s": nbytes / 2 / (k2r / 1e3) / 1e6,
                **welch_ref_traffic(nbytes),
                "stopwatch_around_gj_welch_dev_ms": ref_point["k2_ms"],
                "parity_on_sample": None if par is None else par["psd_1024"]}
            ab = ref_point["ab"]

            def ab_of(n):
                ks, fs_ = [k for k, _ in ab[n]], [f for _, f in ab[n]]
                return {"kernel_ms_avg": sum(ks) / len(ks), "kernel_ms_min": min(ks), "finalize_ms_avg": sum(fs_) / len(fs_),
                        "kernel_plus_finalize_ms_avg": (sum(ks) + sum(fs_)) / len(ks)}
            a4096, a1024 = ab_of(NPERSEG), ab_of(REF_NPERSEG)
            v4096, v1024 = pmc_valu("profiles/r06_pmc_welch/summary.json"), pmc_valu("profiles/r06_pmc_welch1024/summary.json")
            line["secondary"]["K2 at nperseg 4096 and 1024, interleaved A/B on this box, kernel and finalize apart"] = {
                "what": "six rounds of (4096, 1024) back to back on one stream with nothing else in flight; HIP events around "
                        "welch_kernel<N> and around welch_finalize_kernel (gj_welch_timed_dev)",
                "nperseg_4096": a4096, "nperseg_1024": a1024,
                "kernel_ratio_1024_over_4096": a1024["kernel_ms_avg"] / a4096["kernel_ms_avg"],
                "valu_instruction_ratio_1024_over_4096": (v1024 / v4096) if (v4096 and v1024) else None,
                "note": "the 'solo' figures elsewhere in this line are events around gj_welch_dev as a whole (transform + finalize "
                        "+ the gap between the two launches), each from its own place in the run; these are the ones to compare"}
            dep = ref_point["deployment"]["line"]
            if par is not None:
                dep["parity_vs_oracle"] = par["deployment"]
            line["secondary"]["deployment at the reference's sizes: 3 antennas x 10 s (40.96 MB each), scan + K2 at 1024 + "
                              "K5 at N = 50 000 (triangulateTDOA.py:26), 3 pairs"] = dep
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if per_rank is not None:
            line["per_rank"] = per_rank
        if world == 1 and not args.no_end_to_end:
            try:                        # a secondary figure must never cost the primary line
                line["end_to_end"] = end_to_end(np, dev, cap, nbytes)
            except Exception as e:      # e.g. no room for the scratch copy of the capture
                line["end_to_end"] = {"error": repr(e)}
        emit(line)
        # a run that replaces one whose RCCL group could not form counts only when it is right and really had a GPU per rank
        if G.fallback_of and not (line["self_check"]["passed"] and (line["distinct_devices"] == world or args.share_gpu)):
            status = 1
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    stream.close()
    dev.close()
    if rank == 0 and status:
        raise SystemExit(status)


def run_split(args, np, torch, gpsjam, G, dev, work_stream, world, rank):
    """--split: the SAME total work at every N (strong scaling) -- `antennas` captures laid end to end and cut into
    N runs of 2-s units, one run per GPU (gpsjam.split): every GPU scans and transforms its run, one all-gather
    carries the parts' TDOA slots, the antenna pairs are dealt over the ranks, one gather brings the part vectors to
    rank 0, which rebuilds every capture's arrays and runs the tail kernels of a single-GPU stream.  Results are
    bit-identical to the unsplit run (tests/test_split_gpu.py)."""
    from gpsjam import split
    from gpsjam.synth import StreamSpec
    dist = G.dist
    nbytes, A = args.capture_bytes, args.antennas
    nsamp = nbytes // 2
    specs = [stream_spec(StreamSpec, a, nsamp) for a in range(A)]

    def make_buffer(part, b0, b1):
        t = torch.zeros(b1 - b0, dtype=torch.uint8, device="cuda")
        dev.synth_dev(specs[part.antenna], (b1 - b0) // 2, t, first_sample=b0 // 2)
        return t

    def make_noise(antenna, n):
        t = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dev.synth_dev(specs[antenna], n // 2, t, first_sample=0)
        return t

    emulated = int(args.emulate_world) if (world == 1 and args.emulate_world > 1) else 0
    if emulated:
        # rehearsal: this GPU is rank 0 of `emulated` -- 1/emulated of the bytes, its share of the pairs and the combine over
        # all ranks' part vectors; what the other ranks would send was computed here, once, before the timed region
        st = split.emulated_rank(dev, [nbytes] * A, make_buffer, make_noise, emulated, args.emulate_rank, nperseg=NPERSEG,
                                  chunk_samples=CHUNK_SAMPLES, slice_samples=SLICE, overlap=not args.no_overlap,
                                  exchange_always=args.force_exchange, pack_on_side=not args.pack_on_main, group=G.data)
    else:
        st = split.SplitStreams(dev, [nbytes] * A, make_buffer, make_noise, rank=rank, world_size=world, nperseg=NPERSEG,
                                chunk_samples=CHUNK_SAMPLES, slice_samples=SLICE, overlap=not args.no_overlap,
                                exchange_always=args.force_exchange and world == 1, pack_on_side=not args.pack_on_main, group=G.data)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.precondition + args.warmup):
        st.step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ex = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps if world > 1 else 0)]
    ex_end = getattr(st, "_comb", None) or st._side      # rank 0's combine runs on a third stream
    t0 = time.perf_counter()
    got = None
    for k in range(args.steps):
        st.stream_scan()
        ev[k][0].record()
        st.welch()
        ev[k][1].record()
        if ex:
            ex[k][0].record(st._side)
        st.tdoa()
        got = st.exchange(0)
        if ex:
            ex[k][1].record(ex_end)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = G.reduce(torch, elapsed, "max") if world > 1 else elapsed
    welch_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    per_rank = None
    if world > 1:
        # the split's exchange starts on the second stream and ends on rank 0's combine stream: start-to-end as the host
        # sees the two events (both complete by now), not a single-stream interval
        exchange_ms = sum(max(a.elapsed_time(b), 0.0) for a, b in ex) / max(len(ex), 1)
        per_rank = G.per_rank(torch, world, {"k2_ms": welch_ms, "scan_ms": float("nan"), "exchange_ms": exchange_ms,
                                             "step_ms": local_elapsed / max(args.steps, 1) * 1e3})
    own = sum(p.own_bytes for p in st.mine)
    # the second stream's chain (scan, slots, [all-gather], K5, [gather], combine) against K2, one step in isolation:
    # does the chain end inside K2 or does it stick out?
    chain = None
    if rank == 0 and st.overlap and world == 1 and st.is_root:
        torch.cuda.synchronize()
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        reps, k2_ms, front_ms, tail_ms = 5, 0.0, 0.0, 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            e0.record(work_stream)
            st.stream_scan()
            st.welch()
            e1.record(work_stream)
            st.tdoa()
            e3.record(st._side)
            st.exchange(0)
            e2.record(st._comb)
            torch.cuda.synchronize()
            k2_ms += e0.elapsed_time(e1) / reps
            front_ms += e0.elapsed_time(e3) / reps
            tail_ms += e1.elapsed_time(e2) / reps
        chain = {"k2_ms": k2_ms, "scan_to_k5_ms": front_ms, "pack_gather_combine_ms": tail_ms,
                 "hidden_under_k2": bool(front_ms + tail_ms <= k2_ms), "slack_ms": k2_ms - front_ms - tail_ms,
                 "combine_launches": getattr(st, "combine_launches", None),
                 "note": "one step alone.  scan_to_k5: the second stream's front chain (fused scan, tail kernels, slots, "
                         "[all-gather], pick, K5) from the step's start; pack_gather_combine: from the end of K2 to the end "
                         "of the combine (pack on the first stream, then [gather] + assemble / statistics / pack on the "
                         "second).  In steady state step k's combine and step k+1's front chain share the second stream "
                         "under step k+1's K2: hidden when their sum fits into K2"}
    proof = exchange_proof(args, torch, G, dev, None, world, rank)
    if emulated and args.emulate_rank != 0:
        # a rank that only sends: no results arrive here; its step time is what the rehearsal is after
        emit({"metric": "Msamples/s uint8 I/Q through PSD+TDOA xcorr", "value": float(nsamp) * A * args.steps / elapsed / 1e6,
              "unit": "Msamples/s", "projected": True, "emulated_world": emulated, "emulated_rank": args.emulate_rank,
              "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
              "scaling": "strong", "higher_is_better": True, "dtype": "f32", "data": "synthetic", "vs_baseline": None,
              "config": {"workload": f"REHEARSAL: rank {args.emulate_rank} of {emulated} of the split run (its parts only; sends its slots "
                                     "and part vectors, receives nothing)", "own_bytes": own,
                         "pairs": [list(p) for p in st.pairs]},
              "roofline": {"bound": "hbm", "kernel": "welch_kernel<4096> + welch_finalize_kernel over this rank's parts",
                           "achieved": (own / 1e9) / (welch_ms / 1e3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": (own / 1e9) / (welch_ms / 1e3) / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": welch_ms},
              **proof})
        st.close()
        return
    if rank == 0:
        results, tdoa = got.unpack()
        onsets = [r.onset for r in results]
        total_samples = float(nsamp) * A * args.steps
        achieved = (own / 1e9) / (welch_ms / 1e3) if welch_ms > 0 else 0.0
        line = {
            "metric": "Msamples/s uint8 I/Q through PSD+TDOA xcorr", "value": total_samples / elapsed / 1e6,
            "projected": bool(emulated), "emulated_world": emulated,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "precondition_steps": args.precondition, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{A} antenna captures of {nbytes} bytes each (the reference's deployment: three files, "
                                   "worker.py:586-600) through K1-K5, cut into contiguous runs of 8 192 000-byte units over "
                                   f"{world} GPU(s) (SURVEY 8(e)); all {len(tdoa.pairs)} pairs solved; total work is the same at every N",
                       "capture_bytes": nbytes, "antennas": A, "nperseg": NPERSEG, "chunk_samples": CHUNK_SAMPLES,
                       "xcorr_slice": SLICE, "sharding": "captures split into parts (gpsjam.split)"
                       + (f"; REHEARSAL: this GPU is rank 0 of {emulated} (its parts only, the other ranks' slots and part "
                          f"vectors prepared before the timed region); value = the whole job's samples over rank 0's step "
                          f"time, i.e. what {emulated} GPUs would deliver if the other ranks keep pace and the wire is free"
                          if emulated else ""),
                       "own_bytes_rank0": own, "pairs_rank0": [list(p) for p in st.pairs],
                       "parts": [[p.antenna, p.part, p.parts, p.first_byte, p.own_bytes, p.rank] for p in st.parts],
                       "backend": G.backend if world > 1 else None, "transport": G.label if world > 1 else None},
            **proof,
            **G.line_fields(),
            "forced_exchange": bool(args.force_exchange and world == 1),
            "roofline": {"bound": "hbm", "kernel": "welch_kernel<4096> + welch_finalize_kernel over rank 0's parts",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": own, "avg_launch_ms": welch_ms,
                         "overlap": bool(st.overlap),
                         "note": "K2 over the bytes rank 0 owns, as run beside the scan / TDOA kernels (DESIGN.md section 5)"},
            "results": {"pairs": [list(p) for p in tdoa.pairs], "lags": tdoa.lags, "onsets": onsets,
                        "amp_mean": [r.amp_mean for r in results], "baseline": [r.baseline for r in results]},
            "self_check": self_check(results, tdoa, onsets, nsamp, A, proof=proof, world=world, share_gpu=args.share_gpu),
            "host": host_info(),
        }
        if chain is not None:
            line["second_stream_chain"] = chain
        if per_rank is not None:
            line["per_rank"] = per_rank
        emit(line)
    st.close()


ID_BYTES = 256


def parse_identity(text):
    """'rank=0 pid=1 host=h pci=0000:05:00.0 uuid=ab.. hip=0 torch=..' -> dict (values stay strings but rank / pid / hip)."""
    out = {}
    for tok in text.split():
        k, _, v = tok.partition("=")
        out[k] = int(v) if k in ("rank", "pid", "hip") and v.lstrip("-").isdigit() else v
    return out


def torch_device_identity(torch, index):
    """torch's own view of the device a rank computes on (uuid / PCI ids when this torch exposes them)."""
    try:
        pr = torch.cuda.get_device_properties(index)
    except Exception:
        return "none"
    bits = [str(getattr(pr, "uuid", "?"))]
    if hasattr(pr, "pci_bus_id"):
        bits.append("%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0)))
    return "/".join(bits)


def exchange_proof(args, torch, G, dev, comm, world, rank):
    """What the line claims about the exchange, read from the LIVE communicators instead of the command line, and the
    physical device of every rank, all-gathered as bytes through the same collective path the step uses
    (torch.distributed all-gather, or gj_comm_allgather_dev with --transport rccl).  A collective: every rank calls it.
      rccl_ranks   ranks of the RCCL communicator that carried the exchange (0: none did -- gloo, or no group)
      devices      one record per rank: pid, host, PCI bus id and uuid of the gpsjam context's GPU, torch's view
      rehearsal    true when the ranks did not have a GPU each over RCCL (--share-gpu, gloo, --emulate-world)"""
    import socket
    dist = G.dist
    text = (f"rank={rank} pid={os.getpid()} host={socket.gethostname()} {dev.identity()} "
            f"torch={torch_device_identity(torch, torch.cuda.current_device())}")
    raw = text.encode()[:ID_BYTES].ljust(ID_BYTES, b" ")
    mine = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    rccl_ranks, via = 0, "none (single process, no group)"
    if comm is not None:                                    # the library's own RCCL communicator
        rows = torch.zeros((world, ID_BYTES), dtype=torch.uint8, device="cuda")
        comm.allgather(mine, ID_BYTES, rows)
        comm.dev.synchronize()
        live_rank, rccl_ranks, live_dev = comm.live()
        via = f"gj_comm_allgather_dev (ncclCommCount {rccl_ranks}, ncclCommUserRank {live_rank}, ncclCommCuDevice {live_dev})"
    elif dist is not None and dist.is_initialized():
        n = dist.get_world_size(G.data)
        rows = torch.zeros((n, ID_BYTES), dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(rows.view(-1), mine, group=G.data)       # the group that carried the step's exchange
        backend = dist.get_backend(G.data)
        rccl_ranks = n if backend == "nccl" else 0
        via = f"torch.distributed.all_gather_into_tensor over {backend} (get_world_size {n})"
    else:
        rows = mine.unsqueeze(0)
    torch.cuda.synchronize()
    devices = [parse_identity(bytes(r.tolist()).decode(errors="replace")) for r in rows.cpu()]
    distinct = len({(d.get("host"), d.get("pci"), d.get("uuid")) for d in devices})
    rehearsal = bool(args.share_gpu or args.emulate_world or (world > 1 and rccl_ranks != world))
    return {"rccl_ranks": rccl_ranks, "ranks_seen": len(devices), "devices": devices, "distinct_devices": distinct,
            "rehearsal": rehearsal, "identity_exchanged_via": via}


class Groups:
    """The process groups of one run.  `dist`: torch.distributed or None (single process, no group); `data`: the group
    that carries the step's exchange (None = the default group); `backend` / `label`: what that group is made of."""

    def __init__(self, dist=None, data=None, backend=None, label=None, fallback_of=None, probe=None):
        self.dist, self.data, self.backend, self.label, self.fallback_of, self.probe = dist, data, backend, label, fallback_of, probe

    def _device(self):
        return "cuda" if self.dist.get_backend() == "nccl" else "cpu"       # of the DEFAULT group: control traffic

    def reduce(self, torch, value, op):
        """max / sum of one number over the ranks (control traffic: the default group)."""
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return float(t.item())

    def per_rank(self, torch, world, figures):
        """One small all-gather after the timed region: every rank's own figures, so that the first N > 1 curve can be read
        rank by rank (which rank's K2 was slow, whose exchange stuck out) and not only as the max over ranks."""
        names = sorted(figures)
        mine = torch.tensor([float(figures[k]) for k in names], dtype=torch.float64, device=self._device())
        rows = torch.zeros(world * len(names), dtype=torch.float64, device=self._device())
        self.dist.all_gather_into_tensor(rows, mine)
        rows = rows.view(world, len(names)).cpu()
        out = {}
        for j, k in enumerate(names):
            col = [float(v) for v in rows[:, j].tolist()]
            good = [v for v in col if v == v]
            out[k] = {"min": min(good) if good else None, "max": max(good) if good else None, "by_rank": [v if v == v else None for v in col]}
        out["what"] = ("k2_ms: average K2 launch as run (HIP events on its stream); scan_ms: the fused scan + tail alone, after the "
                       "timed region; exchange_ms: second-stream events around slot -> slot all-gather -> K5 -> pack -> result "
                       "gather of every step; step_ms: the rank's own wall time per step before it waits for the others")
        return out

    def line_fields(self):
        if self.dist is None:
            return {}
        out = {"exchange_backend": self.backend, "control_backend": self.dist.get_backend()}
        if self.probe is not None:
            out["rccl_probe"] = self.probe
        if self.fallback_of:
            out["fallback"] = True
            out["fallback_of"] = self.fallback_of
        return out


def rccl_probe(args, dist, world, rank, local_rank):
    """Can N ranks form an RCCL group on this node and move a slot-sized message through it?  Asked in CHILD processes,
    one per rank, before this process has touched the GPU: a rendezvous over a port of its own, an all-reduce, an
    all-gather of 1 MiB per rank (`--rendezvous-only --probe-bytes`).  A child that raises, aborts (RCCL's watchdog
    ends a process whose collective timed out) or hangs past --probe-timeout costs the child, not the rank -- the
    ranks then agree over the control group and carry on over gloo, so that the first N > 1 run on a node yields a
    labelled line whatever RCCL does (VERDICT r05 "next" 2).  Returns (ok on every rank, text)."""
    import socket
    import subprocess
    port = [0]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k != "TORCH_NCCL_ASYNC_ERROR_HANDLING"}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=str(0 if args.share_gpu else local_rank))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--rendezvous-only", "--backend", "nccl",
           "--no-diagnosis", "--probe-bytes", str(1 << 20)]
    if args.share_gpu:
        cmd.append("--share-gpu")
    if args.inject_probe_failure:
        cmd.append("--probe-fail")
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.probe_timeout)
        rc, tail = r.returncode, (r.stderr or "").strip().splitlines()[-3:]
    except subprocess.TimeoutExpired:
        rc, tail = 124, [f"no answer within {args.probe_timeout:.0f} s"]
    mine = {"rank": rank, "status": rc, "seconds": round(time.perf_counter() - t0, 1), "said": " | ".join(tail)[-300:] if rc else ""}
    rows = [None] * world
    dist.all_gather_object(rows, mine)
    bad = [r for r in rows if r["status"] != 0]
    ok = not bad
    text = ("every rank's child formed the group and all-gathered 1 MiB" if ok else
            f"RCCL probe failed on rank(s) {[r['rank'] for r in bad]}: status {bad[0]['status']}: {bad[0]['said']}")
    print(f"[bench] rank {rank}: RCCL probe {'ok' if ok else 'FAILED'} ({mine['seconds']} s): {text}", file=sys.stderr, flush=True)
    return ok, text, {"ok": ok, "seconds_max": max(r["seconds"] for r in rows), "statuses": [r["status"] for r in rows]}


def form_groups(args, torch, world, rank, local_rank):
    """The run's process groups.  N > 1 over nccl: a gloo group first (control: barriers, votes, the figures gathered
    after the timed region -- it cannot fail on RCCL's account), the RCCL probe in child processes, then the exchange
    group: RCCL when the probe passed on every rank, otherwise the gloo group itself, labelled as a fallback.  Nothing
    here touches the GPU in this process."""
    import datetime
    import torch.distributed as dist
    limit = datetime.timedelta(seconds=300)   # a collective that never completes must end the run, not hang it
    if args.backend == "nccl":
        # one node: RCCL's bootstrap needs no outside interface (the container's hostname may not resolve) and
        # there is no InfiniBand to probe; the data path is xGMI peer-to-peer either way
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    if args.backend != "nccl" or (world == 1 and not args.mixed_groups) or args.no_probe:
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        else:
            dist.init_process_group(backend=args.backend, timeout=limit)
        label = args.backend + (" (fallback)" if args.fallback_of else "")
        return Groups(dist, None, args.backend, label, args.fallback_of)
    dist.init_process_group(backend="gloo", timeout=limit)
    if world == 1:                             # --mixed-groups: one rank has nobody to probe with
        ok, text, probe = True, "", None
    else:
        ok, text, probe = rccl_probe(args, dist, world, rank, local_rank)
    if not ok:
        return Groups(dist, None, "gloo", "gloo (fallback)", text, probe)
    torch.cuda.set_device(local_rank)
    try:
        data = dist.new_group(backend="nccl", timeout=limit, device_id=torch.device("cuda", local_rank))
    except TypeError:                          # a torch whose new_group does not take device_id
        data = dist.new_group(backend="nccl", timeout=limit)
    return Groups(dist, data, "nccl", "nccl", None, probe)


def groups_only(args, torch, G, world, rank):
    """--groups-only: the groups a run would use, one small row all-gathered over the exchange group, and the verdict as
    rank 0's line.  Needs no GPU when the exchange group is gloo (which is what a machine without one ends up with)."""
    dist = G.dist
    if dist is None:
        emit({"groups": "single process", "world": 1})
        return 0
    on_gpu = G.backend == "nccl"
    mine = torch.full((64,), rank, dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
    rows = torch.zeros(world * 64, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(rows, mine, group=G.data)
    ok = rows.view(world, 64)[:, 0].cpu().tolist() == list(range(world))
    per_rank = G.per_rank(torch, world, {"k2_ms": 1.0 + rank, "scan_ms": 0.25, "exchange_ms": 0.5 * (rank + 1), "step_ms": 2.0})
    backend = dist.get_backend(G.data)
    if rank == 0:
        emit({"groups": "ok" if ok else "wrong rows", "world": world, "transport": G.label,
              "rccl_ranks": world if backend == "nccl" else 0, "per_rank": per_rank, **G.line_fields()})
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def run_ranks(n, argv, limit_s, stdout=None):
    """N ranks of this script under torch.distributed.run as ONE child process; returns (status, its stdout or None).
    124 = stopped at the limit."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:                  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] launching " + " ".join(cmd), file=sys.stderr, flush=True)
    # same session and process group as this parent: whatever stops the parent's group stops the ranks too; a
    # SIGTERM / SIGINT sent to the parent alone is handed on (torch.distributed.run takes its workers down on SIGTERM)
    proc = subprocess.Popen(cmd, env=env, stdout=stdout, text=True if stdout is not None else None)

    def hand_on(signum, _frame):
        if proc.poll() is None:
            proc.send_signal(signal.SIGTERM)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, hand_on)
    try:
        out, _ = proc.communicate(timeout=limit_s)
        return proc.returncode, out
    except subprocess.TimeoutExpired:
        print(f"[bench] ranks still running after {limit_s:.0f} s: stopping them", file=sys.stderr, flush=True)
    proc.terminate()                            # exactly the child this call started
    try:
        out, _ = proc.communicate(timeout=30)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    return 124, out


def launch_ranks(n, argv, limit_s, args=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under
    torch.distributed.run as a CHILD process (never an exec; this parent has not initialised the GPU
    and never does), pass their output through -- rank 0's JSON line goes to stdout as it is -- and
    return their exit status: non-zero when any rank died, when the 300-s collective timeout inside
    the ranks fired, or when the whole run outlived --launch-timeout.
    A failed run diagnoses itself: ONE fresh `--rendezvous-only` child with the same rank count, backend and device
    assignment follows, and its verdict goes to stderr -- "the ranks cannot form a group on this node" and "the group
    forms, the failure is in the step" are told apart without a second manual run.  The status returned is the first
    run's."""
    import subprocess
    # rank 0's line is held back until the run's status is known: a failed first run must not leave a line in front of
    # the fallback run's
    status, out = run_ranks(n, argv, limit_s, stdout=subprocess.PIPE)
    if status == 0 or args is None or args.no_diagnosis:
        sys.stdout.write(out or "")
        sys.stdout.flush()
        return status
    if out:
        print("[bench] output of the failed run (not passed on):\n" + out, file=sys.stderr, flush=True)
    probe = ["--gpus", str(n), "--backend", args.backend, "--rendezvous-only", "--no-diagnosis"]
    if args.share_gpu:
        probe.append("--share-gpu")
    print(f"[bench] the ranks ended with status {status}; diagnosis: one fresh --rendezvous-only run of {n} ranks "
          f"over {args.backend}", file=sys.stderr, flush=True)
    d_status, out = run_ranks(n, probe, min(240.0, limit_s), stdout=subprocess.PIPE)
    line = next((ln for ln in (out or "").splitlines() if ln.startswith("{")), None)
    if d_status == 0 and line:
        print(f"[bench] diagnosis: rendezvous ok -- {n} ranks formed a {args.backend} group and all-reduced: {line}\n"
              f"[bench] diagnosis: the failure above is in the run itself (a rank died or a collective of the step "
              f"never completed), not in forming the group", file=sys.stderr, flush=True)
    else:
        print(f"[bench] diagnosis: rendezvous FAILED too (status {d_status}"
              + (f", said {line}" if line else ", no line from rank 0") + f"): {n} ranks cannot form a {args.backend} "
              f"group on this node -- look at the launcher / RCCL messages above before anything in the DSP path",
              file=sys.stderr, flush=True)
    if args.backend != "nccl" or args.rendezvous_only:
        return status
    # Second line of defence (the ranks probe RCCL themselves before they use it: rccl_probe): the run over nccl ended
    # non-zero all the same.  ONE more fresh child, every rank on its own GPU, the exchange host-staged over gloo; its
    # line says so (transport "gloo (fallback)", rccl_ranks 0, fallback_of) and its status is 0 only if its self_check
    # passes with a GPU per rank.  The per-rank compute -- what the scaling target measures; the exchange is <= 1 MiB
    # + 164 KB per rank -- does not need RCCL to be timed.
    why = (f"the run over nccl ended with status {status}; rendezvous-only diagnosis: "
           + ("the group forms" if (d_status == 0 and line) else f"the group cannot form (status {d_status})"))
    again, skip = [], 0
    for a in argv:                              # the same command line, --backend replaced
        if skip:
            skip -= 1
        elif a == "--backend":
            skip = 1
        elif not a.startswith("--backend="):
            again.append(a)
    again += ["--backend", "gloo", "--fallback-of", why, "--no-diagnosis"]
    print(f"[bench] fallback: one fresh run of {n} ranks over gloo ({why})", file=sys.stderr, flush=True)
    f_status, f_out = run_ranks(n, again, limit_s, stdout=subprocess.PIPE)
    sys.stdout.write(f_out or "")
    sys.stdout.flush()
    return 0 if f_status == 0 else status


def rendezvous_only(args, torch, world, rank, local_rank):
    """Form the process group, all-reduce one number, rank 0 prints what it saw: separates "the ranks
    cannot find each other / RCCL cannot start" from anything the DSP path does."""
    import datetime
    if args.probe_fail:                         # --inject-probe-failure: as if RCCL could not form the group
        print(f"[bench] probe rank {rank}: --probe-fail, leaving with status 3", file=sys.stderr, flush=True)
        return 3
    if world == 1:
        emit({"rendezvous": "single process", "world": 1})
        return 0
    import torch.distributed as dist
    limit = datetime.timedelta(seconds=120)
    if args.backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        t = torch.tensor([float(rank + 1)], device="cuda")
    else:
        dist.init_process_group(backend=args.backend, timeout=limit)
        t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    ok = float(t.item()) == world * (world + 1) / 2
    # who sat where: every rank's pid and (with a GPU) torch's view of its device, all-gathered over the group just made
    ident = f"rank={rank} pid={os.getpid()} torch=" + (torch_device_identity(torch, local_rank) if t.is_cuda else "cpu")
    mine = torch.frombuffer(bytearray(ident.encode()[:ID_BYTES].ljust(ID_BYTES, b" ")), dtype=torch.uint8).to(t.device)
    rows = torch.zeros((world, ID_BYTES), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(rows.view(-1), mine)
    seats = [parse_identity(bytes(r.tolist()).decode(errors="replace")) for r in rows.cpu()]
    ok = ok and [d.get("rank") for d in seats] == list(range(world))
    if args.probe_bytes > 0:                    # a message of the slot exchange's size through the group's data path
        big = torch.full((args.probe_bytes,), rank % 251, dtype=torch.uint8, device=t.device)
        got = torch.zeros(world * args.probe_bytes, dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(got, big)
        ok = ok and got.view(world, -1)[:, -1].cpu().tolist() == [r % 251 for r in range(world)]
    if rank == 0:
        emit({"rendezvous": "ok" if ok else "wrong sum", "world": dist.get_world_size(), "backend": dist.get_backend(),
              "sum": float(t.item()), "seats": seats})
    if args.fail_rank == rank:
        print(f"[bench] rank {rank}: --fail-rank, leaving with status 3", file=sys.stderr, flush=True)
        os._exit(3)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def self_check(results, tdoa, onsets, nsamp, n_ant, n_pairs=None, proof=None, world=1, share_gpu=False):
    """The synthetic captures carry a known answer: antenna a sees the common burst DELAYS[a]
    samples late, so for every solved pair  lag(i, j) + onset_j - onset_i == DELAYS[j] - DELAYS[i];
    every stream's jammed byte range must be the burst span (to one 64-KiB chunk).  N > 1 ranks must have reported N
    distinct physical devices (unless --share-gpu said they would not): N ranks on one GPU is not an N-GPU run."""
    ok_pairs = []
    for (i, j), lag in zip(tdoa.pairs, tdoa.lags):
        ok_pairs.append(lag + onsets[j] - onsets[i] == DELAYS[j % len(DELAYS)] - DELAYS[i % len(DELAYS)])
    ok_ranges, ok_rank = [], []
    for k, r in enumerate(results):
        rng = r.jamming_byte_ranges()
        want0 = 2 * (int(JAM_SPAN[0] * nsamp) + DELAYS[r.rank % len(DELAYS)])
        want1 = 2 * (int(JAM_SPAN[1] * nsamp) + DELAYS[r.rank % len(DELAYS)])
        ok_ranges.append(len(rng) == 1 and abs(rng[0][0] - want0) <= 65536 and abs(rng[0][1] - want1) <= 65536)
        ok_rank.append(r.rank == k and r.amp_first == 0 and r.amp_count == nsamp and r.onset > 0)
    if n_pairs is None:
        n_pairs = n_ant * (n_ant - 1) // 2
    devices_ok = True
    if proof is not None:
        seen = [d.get("rank") for d in proof["devices"]]
        devices_ok = seen == list(range(world)) and (world == 1 or share_gpu or proof["distinct_devices"] == world)
    return {"tdoa_pairs_ok": bool(ok_pairs) and all(ok_pairs) and len(ok_pairs) == n_pairs,
            "pairs_checked": len(ok_pairs), "jamming_ranges_ok": all(ok_ranges), "streams_ok": all(ok_rank),
            "streams_checked": len(results), "devices_ok": devices_ok,
            "passed": bool(ok_pairs) and all(ok_pairs) and len(ok_pairs) == n_pairs and all(ok_ranges) and all(ok_rank)
                      and devices_ok}


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {"cpu_model": model, "cpu_count": os.cpu_count(), "cpus_usable": usable}


def source_hash():
    """sha256 over the K2 sources: ties a committed PMC summary to the kernel it was measured on."""
    h = hashlib.sha256()
    for name in ("k_welch.hip", "fft_core.h"):
        with open(os.path.join(REPO, "gps-jamming_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_summary(nbytes):
    """HBM bytes and VALU instructions per K2 launch from the committed PMC passes (rocprofv3 --pmc
    in separate runs, gfx950 FETCH_SIZE correction calibrated on K2's own access pattern:
    tools/pmc_welch.sh, tools/pmc_summarize.py).  PMC collection cannot run inside the timed bench,
    so the figures are the recorded ones, offered only for the capture size they were measured on, and
    flagged when the kernel sources have changed since."""
    out = {"traffic": None, "valu_insts": None, "source": "no PMC summary found", "matches_build": None}
    for rel in ("profiles/r06_pmc_welch/summary.json", "profiles/r05_pmc_welch/summary.json", "profiles/r04_pmc_welch/summary.json"):
        try:
            with open(os.path.join(REPO, rel)) as f:
                js = json.load(f)
            corr = js["_hbm_bytes_corrected"]
        except (OSError, KeyError, ValueError):
            continue
        if nbytes != CAPTURE_BYTES:
            out["source"] = "PMC summary is for the 1-GiB capture only"
            return out
        out["traffic"] = float(corr["hbm_bytes_per_launch"])
        out["valu_insts"] = js.get("_valu", {}).get("sq_insts_valu_per_launch")
        out["source"] = rel + " (welch_kernel<4096> only; commit " + str(js.get("_commit", "not recorded")) + ")"
        out["matches_build"] = (js.get("_source_hash") == source_hash()) if js.get("_source_hash") else None
        return out
    return out


def pmc_valu(rel):
    """VALU wave-instructions per K2 launch from a committed PMC summary (None when it is not there)."""
    try:
        with open(os.path.join(REPO, rel)) as f:
            return json.load(f).get("_valu", {}).get("sq_insts_valu_per_launch")
    except (OSError, ValueError):
        return None


def family_traffic(rel, nbytes):
    """HBM bytes per repetition of a secondary kernel family from its committed PMC summary
    (tools/pmc_secondary.sh + tools/pmc_family.py: FETCH_SIZE x gfx950 read correction + WRITE_SIZE, summed over
    the family's kernels), with the check that the family's sources still hash to what was measured."""
    out = {"traffic": None, "traffic_unit": "HBM bytes per repetition (all kernels of the family)",
           "traffic_source": "no PMC summary found", "traffic_measured_on_this_source": None}
    try:
        with open(os.path.join(REPO, rel)) as f:
            js = json.load(f)
        corr = js["_hbm_bytes_corrected"]
    except (OSError, KeyError, ValueError):
        return out
    if nbytes != CAPTURE_BYTES:
        out["traffic_source"] = "PMC summary is for the 1-GiB capture only"
        return out
    h = hashlib.sha256()
    for name in js.get("_sources", []):
        with open(os.path.join(REPO, "gps-jamming_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    out["traffic"] = corr.get("hbm_bytes_per_repetition")
    out["traffic_read"] = corr.get("read_bytes")
    out["traffic_write"] = corr.get("write_bytes")
    out["valu_insts"] = js.get("per_repetition", {}).get("SQ_INSTS_VALU")
    out["traffic_source"] = rel + " (commit " + str(js.get("_commit", "not recorded")) + ")"
    out["traffic_measured_on_this_source"] = js.get("_source_hash") == h.hexdigest()[:16]
    return out


def flops_roofline(nbytes, welch_ms, solo_ms):
    """K2 priced in arithmetic: 5 N log2 N flops per 4096-point segment (the butterflies alone; window, detrend
    and |X|^2 not counted) against the FP32 vector peak of the chip (= its FP32 MFMA peak: 256 CUs x 4 SIMDs x
    16 lanes x 2 (packed) x 2 (FMA) x 2.4 GHz).  Butterflies are add-dominated, so an FFT cannot reach the FMA
    peak: the VALU-issue view (roofline_valu) is the binding one."""
    nsamp = nbytes // 2
    full, rem = divmod(nsamp, CHUNK_SAMPLES)
    segs = full * ((CHUNK_SAMPLES - NPERSEG) // (NPERSEG // 2) + 1)
    if rem >= NPERSEG:
        segs += (rem - NPERSEG) // (NPERSEG // 2) + 1
    flops = segs * 5.0 * NPERSEG * 12
    peak = 157.3
    return {"bound": "valu_fp32", "flops_per_launch": flops, "segments_per_launch": segs, "peak": peak, "unit": "TFLOP/s",
            "achieved": flops / (welch_ms / 1e3) / 1e12, "frac": flops / (welch_ms / 1e3) / 1e12 / peak,
            "solo": {"achieved": flops / (solo_ms / 1e3) / 1e12, "frac": flops / (solo_ms / 1e3) / 1e12 / peak},
            "hbm_frac_if_arithmetic_ran_at_peak": (nbytes / 1e9) / (flops / (peak * 1e12)) / HBM_PEAK_GBS,
            "note": "FP32 vector peak = FP32 MFMA peak on gfx950, so no matrix-pipe route lifts this ceiling "
                    "(DESIGN.md section 5)"}


def valu_roofline(pmc, welch_ms, solo_ms):
    """K2 against what binds it: wave-instructions issued per second over the packed-f32 issue
    ceiling measured on this chip (tools/ubench_valu.hip: 504-570 G wave-instr/s for v_pk_*_f32,
    profiles/r01_ubench_valu_lds.txt)."""
    if not pmc.get("valu_insts"):
        return {"bound": "valu_issue", "achieved": None, "note": "no SQ_INSTS_VALU figure in the PMC summary"}
    ceiling = (504e9, 570e9)
    ach = pmc["valu_insts"] / (welch_ms / 1e3)
    solo = pmc["valu_insts"] / (solo_ms / 1e3)
    return {"bound": "valu_issue", "unit": "wave-instr/s", "achieved": ach, "achieved_solo": solo,
            "peak_range": list(ceiling), "frac_range": [ach / ceiling[1], ach / ceiling[0]],
            "frac_range_solo": [solo / ceiling[1], solo / ceiling[0]],
            "inputs": {"sq_insts_valu_per_launch": pmc["valu_insts"], "avg_launch_ms": welch_ms,
                       "solo_launch_ms": solo_ms, "pmc_source": pmc["source"],
                       "ceiling_source": "profiles/r01_ubench_valu_lds.txt (packed f32 issue, all CUs)"}}


def welch_ref_traffic(nbytes):
    """HBM bytes / VALU instructions per welch_kernel<1024> launch from its committed PMC summary
    (tools/pmc_welch.sh <dir> 1024 + tools/pmc_summarize.py), stamped like the 4096-point one."""
    out = {"traffic": None, "traffic_unit": "HBM bytes per launch", "traffic_source": "no PMC summary found",
           "traffic_measured_on_this_source": None}
    rel = "profiles/r06_pmc_welch1024/summary.json"
    try:
        with open(os.path.join(REPO, rel)) as f:
            js = json.load(f)
        corr = js["_hbm_bytes_corrected"]
    except (OSError, KeyError, ValueError):
        return out
    if nbytes != CAPTURE_BYTES:
        out["traffic_source"] = "PMC summary is for the 1-GiB capture only"
        return out
    out["traffic"] = corr.get("hbm_bytes_per_launch")
    out["valu_insts"] = js.get("_valu", {}).get("sq_insts_valu_per_launch")
    out["traffic_source"] = rel + " (commit " + str(js.get("_commit", "not recorded")) + ")"
    out["traffic_measured_on_this_source"] = (js.get("_source_hash") == source_hash()) if js.get("_source_hash") else None
    return out


def deployment(np, torch, gpsjam, dev, StreamSpec, reps=20, graph=True, **local_kw):
    """The reference's deployment at the reference's sizes (worker.py:184-196,586-600): THREE antenna
    captures of 10 s (40 960 000 bytes each).  Two figures:
      resident_step_ms   captures in HBM; per capture the fused scan (K1 power map + K3 + K4) + noise-floor threshold + K2 at
                         nperseg 1024 + TDOA slot, then K5 over the three slots at N = 50 000 (FFT length 131 072), 3 pairs, one
                         result vector per antenna -- gpsjam.local.LocalAntennas: K2 on the main stream, each capture's chain on a
                         side stream of its own, nothing synchronises the host inside a step;
      file_to_results_ms three capture FILES -> everything on the host: gj_ingest_file per file (the kernels run on the pieces
                         as they land), then K5 on the resident captures -- what a user of the drop-ins waits for.
    Returns the line entry and the GPU results (for the parity check against the oracle on the same bytes)."""
    import tempfile
    nb, ns = REF_CAPTURE_BYTES, REF_CAPTURE_BYTES // 2
    pairs = [(0, 1), (0, 2), (1, 2)]
    caps = []
    for a in range(3):
        c = torch.empty(nb, dtype=torch.uint8, device="cuda")
        dev.synth_dev(stream_spec(StreamSpec, a, ns), ns, c)
        caps.append(c)
    nch, rows = dev.chunk_count(nb, 65536), dev.welch_rows(nb, CHUNK_SAMPLES, REF_NPERSEG)
    sb = dev.tdoa_slot_bytes(REF_SLICE)
    from gpsjam.local import LocalAntennas
    # every antenna of the deployment on this GPU: K2 on the main stream, each capture's scan -> threshold -> slot chain on
    # a side stream of its own, K5 over the three slots, one result vector per antenna (gpsjam/local.py)
    st = LocalAntennas(dev, caps, nperseg=REF_NPERSEG, chunk_samples=CHUNK_SAMPLES, slice_samples=REF_SLICE, rssi_threshold=0.0,
                       graph=graph, **local_kw)
    step = st.step
    power, stats, amp, onset, psd, lags = st.power, st.stats, st.amp, st.onset, st.psd, st.lags

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    resident_ms = (time.perf_counter() - t0) / reps * 1e3
    one = []
    for _ in range(5):                  # the latency of ONE step (host waits for it), not the rate of many
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one.append((time.perf_counter() - t0) * 1e3)
    gpu = {"power": [p.cpu().numpy() for p in power], "stats": [x.cpu().numpy() for x in stats],
           "amp_mean": [float(x[3:4].view(torch.float32)[0]) for x in amp], "onset": [int(x[0]) for x in onset],
           "psd0": psd[0].cpu().numpy(), "lags": lags.cpu().tolist(), "host": [c.cpu().numpy() for c in caps]}

    # file -> results
    d = tempfile.gettempdir()
    if os.path.isdir("/dev/shm") and not os.access(d, os.W_OK):
        d = "/dev/shm"
    paths = [os.path.join(d, f"gpsjam_bench_{os.getpid()}_ant{a}.bin") for a in range(3)]
    file_ms, file_seq_ms, same = [], [], None
    try:
        for a in range(3):
            gpu["host"][a].tofile(paths[a])

        def from_files(together=True):
            t0 = time.perf_counter()
            if together:     # gj_ingest_files: the three recordings side by side, a library thread and lane each
                held = dev.ingest_many(paths, rssi_threshold=0.0, welch=(CHUNK_SAMPLES, REF_NPERSEG), want_db=False)
            else:            # one after the other, as the reference reads them
                held = [dev.ingest(pth, rssi_threshold=0.0, welch=(CHUNK_SAMPLES, REF_NPERSEG), want_db=False) for pth in paths]
            pm = [dev.chunk_power(c) for c in held]
            ps = [dev.welch(c, chunk_samples=CHUNK_SAMPLES, nperseg=REF_NPERSEG, want_db=False)[0] for c in held]
            am = [dev.amp_stats(c, 0.0) for c in held]
            on = [dev.onset(c).start_index for c in held]
            lg, _ = dev.xcorr_lags_at(held, on, REF_SLICE, pairs)
            ms = (time.perf_counter() - t0) * 1e3
            for c in held:
                c.free()
            return ms, (pm, ps, am, on, lg.tolist())

        from_files()                                           # lanes, pinned buffers, page cache warm
        for _ in range(3):
            ms, got = from_files()
            file_ms.append(ms)
        same = bool(all(np.array_equal(got[0][a], gpu["power"][a]) for a in range(3)) and got[1][0].tobytes() == gpu["psd0"].tobytes()
                    and got[3] == gpu["onset"] and got[4] == gpu["lags"])
        from_files(together=False)
        for _ in range(3):
            ms, got1 = from_files(together=False)
            file_seq_ms.append(ms)
        same = same and bool(all(np.array_equal(got1[0][a], gpu["power"][a]) for a in range(3)) and got1[3] == gpu["onset"] and got1[4] == gpu["lags"])
    finally:
        for pth in paths:
            try:
                os.remove(pth)
            except OSError:
                pass
    st.close()
    total = 3 * ns
    line = {"captures": 3, "capture_bytes": nb, "nperseg": REF_NPERSEG, "xcorr_slice": REF_SLICE, "pairs": [list(p) for p in pairs],
            "resident_step_ms": resident_ms, "resident_step_latency_ms": min(one),
            "resident_msamples_per_s": total / (resident_ms / 1e3) / 1e6,
            "file_to_results_ms": min(file_ms) if file_ms else None, "file_to_results_ms_all": file_ms,
            "file_to_results_one_by_one_ms": min(file_seq_ms) if file_seq_ms else None, "file_to_results_one_by_one_ms_all": file_seq_ms,
            "file_msamples_per_s": (total / (min(file_ms) / 1e3) / 1e6) if file_ms else None,
            "files_where": d, "file_results_identical_to_resident": same,
            "results": {"lags": gpu["lags"], "onsets": gpu["onset"], "amp_mean": gpu["amp_mean"],
                        "baseline": [float(x[0]) for x in gpu["stats"]]},
            "what": "resident_step: back-to-back steps over three captures already in HBM (rate) and one step alone (latency); "
                    "file_to_results: three page-cache-resident files through gj_ingest_files -- side by side, a thread of the library "
                    "and a lane per file, two to three fill threads each (three Python threads doing the same: 5-11 ms, GIL hand-offs: "
                    "profiles/NOTES_r05.md section 8); file_to_results_one_by_one: gj_ingest_file per file, as the reference reads "
                    "them; either way pieces sized to the capture, kernels on what has landed, then K5 on the "
                    "resident captures, wall clock with PCIe, best of three"}
    return {"line": line, "gpu": gpu}


def end_to_end(np, dev, cap, nbytes):
    """File / host buffer -> results, wall clock, PCIe included (never `value`): one upload of the
    capture (pinned bounce buffers, eight fill threads) + scan + threshold + Welch + the D2H of the
    power map and the PSD rows.  The file leg reads a scratch copy of the same capture."""
    import tempfile
    import gpsjam
    host = cap.cpu().numpy()
    out = {}

    def run(source):
        """upload, then the four calls (round 2's order)"""
        t0 = time.perf_counter()
        c = dev.capture(source)
        t1 = time.perf_counter()
        pm = dev.chunk_power(c)
        psd, _ = dev.welch(c, chunk_samples=CHUNK_SAMPLES, nperseg=NPERSEG, want_db=False)
        st = dev.amp_stats(c, 0.0)
        on = dev.onset(c)
        t2 = time.perf_counter()
        c.free()
        assert pm.size and psd.size and st.count and on.start_index
        return t1 - t0, t2 - t0, (pm, psd, st, on)

    def run_overlapped(source):
        """gj_ingest_*: the fused scan and K2 run on the pieces that have landed while the rest uploads"""
        t0 = time.perf_counter()
        c = dev.ingest(source, rssi_threshold=0.0, welch=(CHUNK_SAMPLES, NPERSEG), want_db=False)
        pm = dev.chunk_power(c)
        psd, _ = dev.welch(c, chunk_samples=CHUNK_SAMPLES, nperseg=NPERSEG, want_db=False)
        st = dev.amp_stats(c, 0.0)
        on = dev.onset(c)
        t2 = time.perf_counter()
        up_ms = c.ingest_ms[0]
        c.free()
        return up_ms / 1e3, t2 - t0, (pm, psd, st, on)

    def same(a, b):
        return bool(np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes() and a[2].sum == b[2].sum
                    and a[2].first_index == b[2].first_index and bytes(a[3]) == bytes(b[3]))

    def best(fn, source, n=3):
        """the fastest of n runs (a single wall-clock sample of a 25-ms call varies by +-25 % from run to run)"""
        runs = [fn(source) for _ in range(n)]
        return min(runs, key=lambda r: r[1])

    run(host)                                               # lanes, pinned buffers, workspaces at their final size, code paths warm
    run_overlapped(host)
    up, tot, ref = best(run, host)
    out["host_buffer_upload_then_run"] = {"upload_ms": up * 1e3, "total_ms": tot * 1e3, "upload_GBps": nbytes / up / 1e9,
                                          "msamples_per_s": nbytes / 2 / tot / 1e6}
    up, tot, got = best(run_overlapped, host)
    out["host_buffer"] = {"upload_ms": up * 1e3, "total_ms": tot * 1e3, "upload_GBps": nbytes / up / 1e9,
                          "msamples_per_s": nbytes / 2 / tot / 1e6, "overlapped": True,
                          "identical_to_upload_then_run": same(got, ref)}
    # a disk-backed directory when it has room (captures live on disks; tmpfs pays a page bookkeeping of its own that
    # has nothing to do with this path), /dev/shm otherwise; page-cache resident either way
    import shutil
    d = tempfile.gettempdir()
    try:
        roomy = shutil.disk_usage(d).free > 3 * nbytes
    except OSError:
        roomy = False
    if not roomy and os.path.isdir("/dev/shm"):
        d = "/dev/shm"
    path = os.path.join(d, f"gpsjam_bench_{os.getpid()}.bin")
    try:
        host.tofile(path)
        up, tot, _ = best(run, path)
        out["file_upload_then_run"] = {"upload_ms": up * 1e3, "total_ms": tot * 1e3, "upload_GBps": nbytes / up / 1e9,
                                       "msamples_per_s": nbytes / 2 / tot / 1e6}
        up, tot, got = best(run_overlapped, path)
        out["file"] = {"upload_ms": up * 1e3, "total_ms": tot * 1e3, "upload_GBps": nbytes / up / 1e9,
                       "msamples_per_s": nbytes / 2 / tot / 1e6, "where": d + (" (tmpfs)" if d == "/dev/shm" else " (disk-backed, page-cache resident)"), "overlapped": True,
                       "identical_to_upload_then_run": same(got, ref)}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    out["what"] = ("best of three; one H2D per capture, K1 power map + K2 Welch 4096 + K3 amp stats + K4 onset and the D2H of their results, "
                   "wall clock.  host_buffer / file: gj_ingest_* -- the kernels run on the pieces that have landed (16 MiB each at this size) while the "
                   "rest uploads; *_upload_then_run: gpsjam.Capture, then the four calls (round 2's order)")
    out["uploads"] = gpsjam.Capture.uploads
    return out


def cpu_baseline(np, cap, n_chunks, stream, gathered, world, ref_point=None):
    """The oracle (a numpy/scipy port of the reference path) on a bounded prefix of the same
    capture, single process / single thread per rank, with parity of the GPU results on that
    prefix (rank 0)."""
    from oracle import gpsjam_oracle as orc
    from gpsjam import sharded
    sample_bytes = min(cap.numel(), n_chunks * 2 * CHUNK_SAMPLES)
    raw = cap[:sample_bytes].cpu().numpy()
    ns = sample_bytes // 2
    n = min(SLICE, ns)
    slices = None
    if world == 1 and stream.n_ant > 1:                     # the same three slices K5 correlates
        torch_slots = stream.slots.cpu()
        slices = [orc.tdoa_unpack(sharded.slot_fields(torch_slots[a], SLICE)[2]) for a in range(stream.n_ant)]
    t0 = time.perf_counter()
    pm = orc.chunk_power(raw)
    orc.power_threshold(pm)
    lin, _, _ = orc.widmo_waterfall(raw, nperseg=NPERSEG)
    _, avg = orc.rssi_amp_stats(raw, 0.0)
    z = orc.tdoa_unpack(raw)
    orc.tdoa_onset(z)
    cpu_lags = []
    if slices is not None:
        for i, j in stream.pairs:
            cpu_lags.append(int(orc.xcorr_lag(slices[j], slices[i])[0]))
    else:
        orc.xcorr_lag(z[:n], z[:n])
    dt = time.perf_counter() - t0
    out = {"value": ns / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"first {sample_bytes} bytes ({ns} samples, {lin.shape[0]} 1-s chunks) of the same capture: "
                     f"chunk power + threshold + Welch 4096 + amp stats + onset + "
                     + (f"{len(cpu_lags)} 2^19-sample xcorr pairs" if slices is not None else "one 2^19 xcorr")
                     + f", numpy {np.__version__} single thread, {dt:.2f} s"}
    if gathered is not None:
        # parity of the GPU path on the same bytes (not timed)
        results, tdoa = gathered.unpack()
        psd = stream.psd[:lin.shape[0]].cpu().numpy()
        keep = lin > 1e-12
        out["parity_on_sample"] = {
            "psd_max_rel_err": float(np.max(np.abs(psd[keep] - lin[keep]) / lin[keep])),
            "power_map_max_rel_err": float(np.max(np.abs(results[0].power_map[:pm.size] - pm) / pm))}
        if slices is not None:
            out["parity_on_sample"]["lags_equal"] = cpu_lags == tdoa.lags
    if ref_point is not None:
        # the reference's own operating point against the oracle on the same bytes (not part of the timed sample above)
        lin_r, _, _ = orc.widmo_waterfall(raw, nperseg=REF_NPERSEG)
        got = ref_point["psd"][:lin_r.shape[0]].cpu().numpy()
        keep = lin_r > 1e-12
        par = {"psd_1024": {"rows": int(lin_r.shape[0]),
                            "psd_max_rel_err": float(np.max(np.abs(got[keep] - lin_r[keep]) / lin_r[keep]))}}
        g = ref_point["deployment"]["gpu"]
        t0 = time.perf_counter()
        pm_err, on_cpu, amp_err, zs = [], [], [], []
        for a in range(3):
            h = g["host"][a]
            pm_a = orc.chunk_power(h)
            pm_err.append(float(np.max(np.abs(g["power"][a] - pm_a) / pm_a)))
            z = orc.tdoa_unpack(h)
            on_cpu.append(int(orc.tdoa_onset(z)))
            zs.append(z)
            _, avg = orc.rssi_amp_stats(h, 0.0)
            amp_err.append(abs(g["amp_mean"][a] - float(avg)) / float(avg))
        base = [float(orc.power_threshold(orc.chunk_power(g["host"][a]))[0]) for a in range(3)]
        lin_d, _, _ = orc.widmo_waterfall(g["host"][0], nperseg=REF_NPERSEG)
        keep = lin_d > 1e-12
        lags_cpu = [int(orc.xcorr_lag(zs[j][on_cpu[j]:on_cpu[j] + REF_SLICE], zs[i][on_cpu[i]:on_cpu[i] + REF_SLICE])[0])
                    for i, j in ((0, 1), (0, 2), (1, 2))]
        par["deployment"] = {"power_map_max_rel_err": max(pm_err), "onsets_equal": on_cpu == g["onset"],
                             "lags_equal": lags_cpu == g["lags"], "amp_mean_max_rel_err": max(amp_err),
                             "baseline_max_rel_err": max(abs(float(g["stats"][a][0]) - base[a]) / base[a] for a in range(3)),
                             "psd_1024_max_rel_err_capture0": float(np.max(np.abs(g["psd0"][keep] - lin_d[keep]) / lin_d[keep])),
                             "cpu_oracle_s": time.perf_counter() - t0,
                             "note": "the oracle (numpy/scipy port, one core) on the same three captures; its time is for power maps, "
                                     "onsets, amplitude means, one capture's Welch 1024 and the three 50 000-sample correlations"}
        out["reference_point_parity"] = par
    return out


if __name__ == "__main__":
    main()
