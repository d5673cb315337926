"""One antenna capture per GPU (one process per GPU, torch.distributed over RCCL/xGMI).

The per-capture kernels (K1 power scan, K2 Welch PSD, K3 amplitude statistics, K4 onset)
are independent across captures, so captures are sharded one per rank with no data-path
collective.  The only real exchange of the path is TDOA: every rank correlates its own
onset-aligned slice against the reference antenna's slice (rank 0), which is broadcast
(2 bytes x slice samples, 1 MiB for 2^19) -- then one gather of a small fixed-layout result
vector (power map, noise floor, threshold, amplitude statistics, onset, lag, peak, mean
spectrum) to rank 0, which runs the host-side solvers (grid search / bearing).

``pack_results`` / ``unpack_results`` / ``exchange`` only touch torch tensors and
torch.distributed, so they run unchanged on CPU tensors with the gloo backend (tests) and on
HIP tensors with the nccl (= RCCL) backend (bench.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import contextlib

import numpy as np
import torch

HEADER = 16          # scalars in front of the vectors, see RESULT_FIELDS
RESULT_FIELDS = ("n_chunks", "baseline", "threshold", "n_above", "amp_first", "amp_count",
                 "amp_mean", "onset", "lag", "peak", "noise_power", "n_rows", "nperseg",
                 "rank", "reserved0", "reserved1")


def result_len(n_chunks: int, nperseg: int) -> int:
    return HEADER + n_chunks + nperseg


def pack_results(n_chunks: int, nperseg: int, power_map: torch.Tensor, stats: torch.Tensor,
                 amp_first: torch.Tensor, amp_count: torch.Tensor, amp_mean: torch.Tensor,
                 onset: torch.Tensor, lag: torch.Tensor, peak: torch.Tensor,
                 noise_power: torch.Tensor, mean_spectrum: torch.Tensor, n_rows: int,
                 rank: int) -> torch.Tensor:
    """float64 vector [HEADER + n_chunks + nperseg] built with device-side ops only (no
    host synchronisation).  int64 scalars are exact in float64 up to 2^53."""
    dev = power_map.device
    head = torch.zeros(HEADER, dtype=torch.float64, device=dev)
    head[0] = n_chunks
    head[1:4] = stats.to(torch.float64)[:3]
    head[4] = amp_first.to(torch.float64)
    head[5] = amp_count.to(torch.float64)
    head[6] = amp_mean.to(torch.float64)
    head[7] = onset.to(torch.float64)
    head[8] = lag.to(torch.float64)
    head[9] = peak.to(torch.float64)
    head[10] = noise_power.to(torch.float64)
    head[11] = n_rows
    head[12] = nperseg
    head[13] = rank
    return torch.cat([head, power_map.to(torch.float64), mean_spectrum.to(torch.float64)])


@dataclass
class StreamResult:
    rank: int
    power_map: np.ndarray
    baseline: float
    threshold: float
    n_above: int
    amp_first: int
    amp_count: int
    amp_mean: float
    onset: int
    lag: int
    peak: float
    noise_power: float
    mean_spectrum: np.ndarray

    def jamming_byte_ranges(self, chunk_bytes: int = 65536):
        """(start_byte, end_byte) runs above the threshold (worker.py:248-264)."""
        mask = self.power_map > np.float32(self.threshold)
        if not mask.any():
            return []
        edges = np.diff(mask.astype(np.int8))
        starts = list(np.where(edges == 1)[0] + 1)
        ends = list(np.where(edges == -1)[0] + 1)
        if mask[0]:
            starts.insert(0, 0)
        if mask[-1]:
            ends.append(mask.size)
        return [(int(s) * chunk_bytes, int(e) * chunk_bytes) for s, e in zip(starts, ends)]


def unpack_results(vec: torch.Tensor) -> StreamResult:
    v = vec.detach().to("cpu", torch.float64).numpy()
    n_chunks, nperseg = int(v[0]), int(v[12])
    pm = v[HEADER:HEADER + n_chunks].astype(np.float32)
    spec = v[HEADER + n_chunks:HEADER + n_chunks + nperseg].astype(np.float32)
    return StreamResult(rank=int(v[13]), power_map=pm, baseline=float(v[1]), threshold=float(v[2]),
                        n_above=int(v[3]), amp_first=int(v[4]), amp_count=int(v[5]),
                        amp_mean=float(v[6]), onset=int(v[7]), lag=int(v[8]), peak=float(v[9]),
                        noise_power=float(v[10]), mean_spectrum=spec)


def broadcast_reference_slice(slice_i16: torch.Tensor, world_size: int, src: int = 0) -> torch.Tensor:
    """Rank ``src``'s onset-aligned slice (int16 view of the I/Q byte pairs) to every rank."""
    if world_size > 1:
        import torch.distributed as dist
        dist.broadcast(slice_i16.view(torch.uint8), src=src)   # neither gloo nor RCCL moves int16
    return slice_i16


def gather_results(vec: torch.Tensor, rank: int, world_size: int, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather every rank's result vector on ``dst``; returns the list there, None elsewhere."""
    if world_size == 1:
        return [vec]
    import torch.distributed as dist
    if rank == dst:
        out = [torch.empty_like(vec) for _ in range(world_size)]
        dist.gather(vec, gather_list=out, dst=dst)
        return out
    dist.gather(vec, gather_list=None, dst=dst)
    return None


class AntennaStream:
    """Device-resident pipeline of one capture on one GPU (uses torch only for device
    memory and the stream; every kernel is a gpsjam C-ABI call)."""

    def __init__(self, dev, capture: torch.Tensor, *, chunk_bytes: int = 65536,
                 chunk_samples: int = 2048000, nperseg: int = 4096, fs: float = 2.048e6,
                 slice_samples: int = 1 << 19, noise_samples: int = 200000, window: int = 1000,
                 factor: float = 50.0, rssi_threshold: float = 0.0, rank: int = 0, world_size: int = 1,
                 overlap: Optional[bool] = None):
        assert capture.dtype == torch.uint8 and capture.is_contiguous()
        self.dev, self.cap = dev, capture
        # K2 is VALU/LDS bound and leaves ~90 % of the HBM bandwidth idle, the fused scan is HBM
        # bound: with ``overlap`` the scan, threshold and TDOA kernels run on a second HIP
        # stream (own gpsjam context = own workspace) concurrently with K2 and join in pack().
        if overlap is None:
            overlap = capture.is_cuda and hasattr(dev, "_ctx")
        self.overlap = bool(overlap)
        self.dev_side = dev
        if capture.is_cuda and hasattr(dev, "_ctx"):
            # the pipeline's torch ops, its events and the gpsjam kernels must share one stream
            self._main = torch.cuda.current_stream(capture.device)
            dev.set_stream(self._main.cuda_stream)
        if self.overlap:
            self.dev_side = type(dev)(dev.index)
            self._side = torch.cuda.Stream(device=capture.device)
            self.dev_side.set_stream(self._side.cuda_stream)
            self._ev_free = torch.cuda.Event()      # main: previous results consumed, buffers may be rewritten
            self._ev_side = torch.cuda.Event()      # side: scan / TDOA results ready
            self._ev_free.record(self._main)
        self.nbytes = capture.numel()
        self.rank, self.world = rank, world_size
        self.chunk_bytes, self.chunk_samples, self.nperseg, self.fs = chunk_bytes, chunk_samples, nperseg, fs
        self.slice_samples, self.noise_samples, self.window, self.factor = slice_samples, noise_samples, window, factor
        self.rssi_threshold = rssi_threshold
        d = capture.device
        self.n_chunks = dev.chunk_count(self.nbytes, chunk_bytes)
        self.rows = dev.welch_rows(self.nbytes, chunk_samples, nperseg)
        self.power = torch.empty(self.n_chunks, dtype=torch.float32, device=d)
        self.stats = torch.empty(3, dtype=torch.float32, device=d)
        self.mask = torch.empty(self.n_chunks, dtype=torch.uint8, device=d)
        self.psd = torch.empty((max(self.rows, 1), nperseg), dtype=torch.float32, device=d)
        self.amp = torch.zeros(4, dtype=torch.int64, device=d)        # gj_amp_stats (32 bytes)
        self.onset = torch.zeros(2, dtype=torch.int64, device=d)      # gj_onset (16 bytes)
        self.starts = torch.zeros(2, dtype=torch.int64, device=d)
        self.lag = torch.zeros(1, dtype=torch.int32, device=d)
        self.peak = torch.zeros(1, dtype=torch.float32, device=d)
        self.ref_slice = torch.zeros(slice_samples, dtype=torch.int16, device=d)
        self._ar = torch.arange(slice_samples, dtype=torch.int64, device=d)
        # two result vectors, used alternately: the gather of step k (second stream) may still be
        # reading one while step k + 1 packs into the other
        self._results = [torch.zeros(result_len(self.n_chunks, nperseg), dtype=torch.float64, device=d) for _ in range(2)]
        self._result_idx = 0
        self.result = self._results[0]
        if self.overlap:
            self._ev_packed = torch.cuda.Event()
        self.cap16 = capture.view(torch.int16)
        ws_side = max(dev.xcorr_workspace(2, slice_samples, 1), self.nbytes // 48 + (1 << 20))
        ws_main = dev.welch_workspace(self.nbytes, chunk_samples, nperseg)
        if self.overlap:
            dev.reserve(ws_main)
            self.dev_side.reserve(ws_side)
        else:
            dev.reserve(max(ws_main, ws_side))

    def _on_side(self):
        """Context manager: torch's current stream = the side stream (no-op without overlap)."""
        return torch.cuda.stream(self._side) if self.overlap else contextlib.nullcontext()

    def stream_scan(self):
        """K1 + K3 + K4 in one pass over the capture, then the noise-floor threshold."""
        if self.overlap:
            self._side.wait_event(self._ev_free)
        d = self.dev_side
        d.stream_scan_dev(self.cap, self.nbytes, self.chunk_bytes, self.power, self.rssi_threshold,
                          self.amp, self.noise_samples, self.window, self.factor, self.onset)
        d.power_threshold_dev(self.power, self.n_chunks, self.stats, self.mask)
        if self.overlap:
            self._ev_side.record(self._side)

    def welch(self):
        self.dev.welch_dev(self.cap, self.nbytes, self.chunk_samples, self.nperseg, self.fs, self.psd)

    def scan(self):
        """Everything that only needs this rank's capture (no host synchronisation)."""
        self.stream_scan()
        self.welch()

    def tdoa(self):
        """Reference slice from rank 0 (broadcast), lag of this capture against it."""
        with self._on_side():
            self._tdoa(self.dev_side)
            if self.overlap:
                self._ev_side.record(self._side)

    def _tdoa(self, dev):
        n = self.slice_samples
        nsamp = self.nbytes // 2
        if self.world > 1:
            if self.rank == 0:
                idx = (self.onset[0] + self._ar).clamp_(0, nsamp - 1)
                torch.index_select(self.cap16, 0, idx, out=self.ref_slice)
            broadcast_reference_slice(self.ref_slice, self.world, 0)
            # rank 0's onset travels in the slice's validity: an un-found onset (-1) on rank 0
            # makes idx start at the clamp and the lag meaningless; rank 0 reports it.
            self.starts[1:2].copy_(self.onset[0:1])          # starts[0] stays 0: the slice is already aligned
            dev.xcorr_lags_dev([self.ref_slice, self.cap], [2 * n, self.nbytes], self.starts, n,
                                    [(0, 1)], self.lag, self.peak)
        else:
            self.starts.copy_(self.onset[0:1].expand(2))
            dev.xcorr_lags_dev([self.cap, self.cap], [self.nbytes, self.nbytes], self.starts, n,
                                    [(0, 1)], self.lag, self.peak)

    def pack(self) -> torch.Tensor:
        """Result vector of this stream, built by one kernel (layout = pack_results)."""
        if self.overlap:
            self._main.wait_event(self._ev_side)
        self._result_idx ^= 1
        self.result = self._results[self._result_idx]
        self.dev.pack_result_dev(self.n_chunks, self.power, self.stats, self.amp, self.onset, self.lag, self.peak,
                                 self.psd, self.rows, self.nperseg, self.rank, self.result)
        if self.overlap:
            self._ev_free.record(self._main)
            self._ev_packed.record(self._main)
        return self.result

    def exchange(self, dst: int = 0) -> Optional[List[torch.Tensor]]:
        """Pack this step's result vector and gather every rank's on ``dst``.  With two streams the
        collective is issued on the second one (after the packing kernel), so the main stream goes
        straight on to the next step's K2 instead of waiting for the gather; the next step's scan
        follows the gather in stream order, and a result buffer is rewritten only two steps later,
        after a join that lies behind it."""
        vec = self.pack()
        if self.world == 1:
            return [vec]
        if not self.overlap:
            return gather_results(vec, self.rank, self.world, dst)
        self._side.wait_event(self._ev_packed)
        with torch.cuda.stream(self._side):
            return gather_results(vec, self.rank, self.world, dst)

    def step(self):
        """One pass of the hot path over this rank's capture + the exchange."""
        self.scan()
        self.tdoa()
        return self.exchange(0)
