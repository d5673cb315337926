#!/usr/bin/env python3
"""Benchmark of the jamming-detection DSP path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic antenna capture per GPU
(BASELINE.json configs[1], one stream per rank = configs[4] for N > 1):
  one fused HBM pass for K1 per-chunk power (+ 5th-percentile/+6 dB threshold), K3 amplitude
  statistics and K4 onset; K2 fused unpack + 4096-pt Welch PSD (1-s chunks); K5 2^20-pt FFT
  cross-correlation of the rank's onset-aligned 2^19-sample slice against the reference
  antenna's slice (rank 0, broadcast over RCCL), then an RCCL gather of the per-stream result
  vector to rank 0.  K2 (VALU/LDS bound) runs on one HIP stream, the HBM-bound scan and the
  TDOA kernels concurrently on a second one; they join before the result is packed.
Captures are generated in HBM before the timed region (2^30 bytes = 536 870 912 I/Q samples
per GPU, integer-only generator, seeds 1234 + rank) -- inputs are resident when timing starts.

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (K2 welch_kernel<4096>
+ its finalize): algorithmic bytes = 2 B x samples per launch, duration from HIP events
recorded on the launch stream around every K2 launch of the timed steps.  ``cpu_baseline``
(N = 1 only) times the numpy/scipy oracle (oracle/gpsjam_oracle.py, kind "port") on a
bounded prefix of the same capture on the host's cores.
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--capture-bytes", type=int, default=CAPTURE_BYTES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo) on a one-GPU box")
    ap.add_argument("--precondition", type=int, default=30,
                    help="untimed steps run before the warm-up to settle clocks (not counted in --warmup)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the scan/TDOA kernels on the K2 stream instead of concurrently on a second one")
    ap.add_argument("--cpu-sample-chunks", type=int, default=24,
                    help="1-s chunks of the capture given to the CPU oracle (bounded sample)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import gpsjam
    from gpsjam.sharded import AntennaStream, unpack_results
    from gpsjam.synth import StreamSpec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    dev = gpsjam.Device(local_rank)
    # one explicit HIP stream for everything in the step: the gpsjam kernels, torch's small
    # packing ops and the events that time the dominant kernel
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    dev.set_stream(work_stream.cuda_stream)
    nbytes = args.capture_bytes
    nsamp = nbytes // 2
    cap = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    spec = StreamSpec(seed=1234, antenna=rank, delay=DELAYS[rank % len(DELAYS)], jam_start=int(0.4 * nsamp),
                      jam_end=int(0.7 * nsamp), noise_sigma=6.25, jam_sigma=60.0 * (1.0, 0.7, 0.5, 0.8)[rank % 4])
    dev.synth_dev(spec, nsamp, cap)
    stream = AntennaStream(dev, cap, nperseg=NPERSEG, chunk_samples=CHUNK_SAMPLES, slice_samples=SLICE,
                           rank=rank, world_size=world, overlap=not args.no_overlap)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
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
    t0 = time.perf_counter()
    gathered = None
    for k in range(args.steps):
        # the step, with HIP events on the launch stream around the dominant kernel
        stream.stream_scan()            # K1 + K3 + K4 in one HBM pass, + noise-floor threshold
        ev[k][0].record()
        stream.welch()                  # K2
        ev[k][1].record()
        stream.tdoa()
        gathered = stream.exchange(0)   # pack (after the join) + RCCL gather, issued on the second stream
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # K2 alone (nothing else in flight), after the timed region: reported next to the as-run
    # figure, which includes whatever the concurrently running scan/TDOA kernels cost it
    solo = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in solo:
        a.record()
        stream.welch()
        b.record()
    torch.cuda.synchronize()
    solo_ms = sum(a.elapsed_time(b) for a, b in solo) / len(solo)

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    welch_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)

    if rank == 0:
        results = [unpack_results(v) for v in gathered]
        total_samples = float(nsamp) * world * args.steps
        value = total_samples / elapsed / 1e6
        achieved = (nbytes / 1e9) / (welch_ms / 1e3) if welch_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(nbytes)
        line = {
            "metric": "Msamples/s uint8 I/Q through PSD+TDOA xcorr",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "precondition_steps": args.precondition,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: fused uint8->complex64 + 4096-pt Welch PSD + jamming power "
                                   "threshold on 1 GiB synthetic I/Q per GPU (+ K3 amp stats, K4 onset, "
                                   "K5 2^20-pt xcorr vs reference antenna, RCCL gather)",
                       "capture_bytes_per_gpu": nbytes, "nperseg": NPERSEG, "chunk_samples": CHUNK_SAMPLES,
                       "xcorr_slice": SLICE, "streams": world, "sharding": "one capture per GPU"},
            "roofline": {"bound": "hbm", "kernel": "welch_kernel<4096> + welch_finalize_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": welch_ms,
                         "overlap": bool(stream.overlap),
                         "solo": {"avg_launch_ms": solo_ms, "achieved": (nbytes / 1e9) / (solo_ms / 1e3),
                                  "frac": (nbytes / 1e9) / (solo_ms / 1e3) / HBM_PEAK_GBS},
                         "note": "K2 is FP32-VALU/LDS bound, not HBM bound (DESIGN.md section 5); "
                                 "'achieved' is K2 as run in the timed steps, i.e. with the HBM-bound "
                                 "scan + TDOA kernels executing concurrently on a second stream when "
                                 "overlap is true; 'solo' is K2 with nothing else in flight"},
            "results": {"lags": [r.lag for r in results], "onsets": [r.onset for r in results],
                        "jamming_ranges_rank0": results[0].jamming_byte_ranges()[:4],
                        "baseline_rank0": results[0].baseline, "amp_mean": [r.amp_mean for r in results]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(np, cap, args.cpu_sample_chunks, stream, results[0])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    dev.close()


def pmc_traffic(nbytes):
    """HBM bytes per K2 launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs, gfx950 correction calibrated on K2's own access pattern: tools/pmc_welch.sh,
    tools/pmc_summarize.py).  PMC collection cannot run inside the timed bench, so the figure is the
    recorded one and only offered for the capture size it was measured on."""
    path = os.path.join(REPO, "profiles", "r01_pmc_welch", "summary_final.json")
    try:
        with open(path) as f:
            corr = json.load(f)["_hbm_bytes_corrected"]
        if nbytes != CAPTURE_BYTES:
            return None, "PMC summary is for the 1-GiB capture only"
        return float(corr["hbm_bytes_per_launch"]), "profiles/r01_pmc_welch/summary_final.json (welch_kernel<4096> only)"
    except (OSError, KeyError, ValueError):
        return None, "no PMC summary found"


def cpu_baseline(np, cap, n_chunks, stream, gpu_result):
    """The oracle (a numpy/scipy port of the reference path) on a bounded prefix of the same
    capture, single process / single thread, with parity of the GPU results on that prefix."""
    from oracle import gpsjam_oracle as orc
    sample_bytes = min(cap.numel(), n_chunks * 2 * CHUNK_SAMPLES)
    raw = cap[:sample_bytes].cpu().numpy()
    ns = sample_bytes // 2
    t0 = time.perf_counter()
    pm = orc.chunk_power(raw)
    orc.power_threshold(pm)
    lin, _, _ = orc.widmo_waterfall(raw, nperseg=NPERSEG)
    _, avg = orc.rssi_amp_stats(raw, 0.0)
    z = orc.tdoa_unpack(raw)
    orc.tdoa_onset(z)
    n = min(SLICE, ns)
    orc.xcorr_lag(z[:n], z[:n])
    dt = time.perf_counter() - t0
    # parity of the GPU path on the same bytes (not timed)
    psd = stream.psd[:lin.shape[0]].cpu().numpy()
    keep = lin > 1e-12
    psd_err = float(np.max(np.abs(psd[keep] - lin[keep]) / lin[keep]))
    pm_err = float(np.max(np.abs(gpu_result.power_map[:pm.size] - pm) / pm))
    return {"value": ns / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"first {sample_bytes} bytes ({ns} samples, {lin.shape[0]} 1-s chunks) of the same capture: "
                      f"chunk power + threshold + Welch 4096 + amp stats + onset + one 2^19 xcorr, "
                      f"numpy {np.__version__} single thread, {dt:.2f} s",
            "parity_on_sample": {"psd_max_rel_err": psd_err, "power_map_max_rel_err": pm_err}}


if __name__ == "__main__":
    main()
