"""Every antenna of a deployment on ONE GPU (one process) -- the reference's own shape: three antenna recordings
handed to one `GPSAnalysisThread` (GpsJammerApp/app/worker.py:97-101,586-600; skrypty/triangulateRSSI.py:147-154,
skrypty/triangulateTDOA.py:60-90 for a pair of them).

`gpsjam.sharded.AntennaStream` is one capture per GPU and `gpsjam.split.SplitStreams` cuts captures over GPUs; this is
the third arrangement, and for captures of the reference's size (10 s = 41 MB) the one that matters: a step over such a
capture is a dozen launches that each take microseconds, so what counts is how many dependent launches stand in a row,
not bytes.  Per step:
  main stream      K2 (Welch PSD) of every capture: ONE launch + one finalize when the captures are of one length
                   (gj_welch_batch_dev), else one after the other -- a K2 launch fills the chip
  side stream a    capture a's fused scan (K1 power map, K3 amplitude statistics, K4 block sums) and its tail (noise-floor
                   threshold, amplitude totals, onset, TDOA slot): two launches (gj_capture_scan_dev; eight until round 5);
                   the captures' chains are independent of each other and run on streams of their own (up to three, each
                   tested to run beside the others: gpsjam/streams.py), so they overlap instead of queueing
  side stream 0    after all slots: K5 over every antenna pair (three launches)
  main stream      one result vector per antenna, one launch for all (gj_pack_results_dev; the layout of
                   gj_pack_result_dev; antenna 0 carries the pair table)
From the second step on the whole step is replayed as ONE captured HIP graph (``graph=True``): a dozen launches of a few
microseconds each (round 4: ~45) are launch-bound when issued one by one -- 0.21 ms replayed, 0.22 eager.
The kernels are those of the other two arrangements; results are byte-equal to `AntennaStream` run on each capture
(tests/test_local_gpu.py).  `step()` returns a `gpsjam.sharded.StepResults`.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

import logging

from .sharded import LAG_INVALID, StepResults, all_pairs, result_len
from .streams import stream_beside_checked

_log = logging.getLogger("gpsjam.local")


class LocalAntennas:
    """``captures``: one contiguous uint8 device tensor (I,Q,I,Q...) per antenna, all on ``dev``'s GPU.  Uses torch only
    for device memory, streams and events; every kernel is a gpsjam C-ABI call and nothing synchronises the host."""

    def __init__(self, dev, captures: Sequence[torch.Tensor], *, chunk_bytes: int = 65536, chunk_samples: int = 2048000,
                 nperseg: int = 1024, fs: float = 2.048e6, slice_samples: int = 50000, noise_samples: int = 200000,
                 window: int = 1000, factor: float = 50.0, rssi_threshold: float = 0.0, side_streams: int = 3,
                 graph: bool = True, scan_first: bool = False):
        assert len(captures) >= 1 and all(c.dtype == torch.uint8 and c.is_contiguous() and c.is_cuda for c in captures)
        self.dev, self.caps = dev, list(captures)
        self.n_ant = len(self.caps)
        d = self.caps[0].device
        self.chunk_bytes, self.chunk_samples, self.nperseg, self.fs = chunk_bytes, chunk_samples, nperseg, fs
        self.slice_samples, self.noise_samples, self.window, self.factor = slice_samples, noise_samples, window, factor
        self.rssi_threshold = rssi_threshold
        self.scan_first = bool(scan_first)
        self._main = torch.cuda.current_stream(d)
        dev.set_stream(self._main.cuda_stream)
        # side streams, each on a hardware queue of its own (tested), each with a context (= workspace) bound to it
        self._sides: List[tuple] = []
        #: True when every side stream was TESTED to run beside the main stream and the other side streams; False: the
        #: step is still correct (cross-stream order is by events) but its chains run one after the other (a warning is logged)
        self.streams_overlap = True
        for _ in range(max(1, min(self.n_ant, int(side_streams)))):
            sdev = type(dev)(dev.index)
            s, ok = stream_beside_checked([(dev, self._main)] + self._sides, device=d)
            self.streams_overlap = self.streams_overlap and ok
            sdev.set_stream(s.cuda_stream)
            self._sides.append((sdev, s))
        self.nbytes = [int(c.numel()) for c in self.caps]
        self.n_chunks = [dev.chunk_count(n, chunk_bytes) for n in self.nbytes]
        self.rows = [dev.welch_rows(n, chunk_samples, nperseg) for n in self.nbytes]
        f32, i64 = torch.float32, torch.int64
        self.power = [torch.empty(max(n, 1), dtype=f32, device=d) for n in self.n_chunks]
        self.stats = [torch.zeros(3, dtype=f32, device=d) for _ in self.caps]
        self.amp = [torch.zeros(4, dtype=i64, device=d) for _ in self.caps]        # gj_amp_stats
        self.onset = [torch.zeros(4, dtype=i64, device=d) for _ in self.caps]      # gj_onset
        self.psd = [torch.empty((max(r, 1), nperseg), dtype=f32, device=d) for r in self.rows]
        self.slot_bytes = dev.tdoa_slot_bytes(slice_samples)
        self.slots = torch.zeros((self.n_ant, self.slot_bytes), dtype=torch.uint8, device=d)
        self.pairs = all_pairs(self.n_ant)
        npairs = max(len(self.pairs), 1)
        self.d_pairs = torch.tensor([x for p in self.pairs for x in p] or [0, 0], dtype=torch.int32, device=d)
        self.lags = torch.full((npairs,), LAG_INVALID, dtype=torch.int32, device=d)
        self.peaks = torch.zeros(npairs, dtype=f32, device=d)
        self.margins = torch.zeros(npairs, dtype=f32, device=d)
        self.final_len = max(result_len(n, nperseg, len(self.pairs)) for n in self.n_chunks)
        self._final = [torch.zeros((self.n_ant, self.final_len), dtype=torch.float64, device=d) for _ in range(2)]
        self._done = [torch.cuda.Event() for _ in range(2)]
        self._idx = 0
        self._ev_go = torch.cuda.Event()
        self._ev_side = [torch.cuda.Event() for _ in self._sides]
        # one captured graph per result set (made at the second / third step); never on the legacy default stream
        self._graphs = [None, None] if (graph and self._main != torch.cuda.default_stream(d)) else None
        self._steps = 0
        self._desc = {}
        # K2 of all captures in one launch when they are of one length and every one has rows (the reference's deployment:
        # three recordings made together); otherwise one launch per capture
        self._k2_batched = (self.n_ant > 1 and len(set(self.nbytes)) == 1 and all(self.rows) and self.n_ant <= 16
                            and all(p.data_ptr() % 16 == 0 for p in self.psd))
        # workspaces: nothing is allocated inside a step
        dev.reserve(max(dev.welch_workspace(n, chunk_samples, nperseg) for n in self.nbytes) * (self.n_ant if self._k2_batched else 1))
        for k, (sdev, _) in enumerate(self._sides):
            ws = max(self.nbytes) // 48 + (1 << 20)
            if k == 0 and self.pairs:
                ws = max(ws, dev.xcorr_workspace(self.n_ant, slice_samples, len(self.pairs)))
            sdev.reserve(ws)

    def _enqueue(self, out: torch.Tensor):
        """The step's launches, origin and end on the main stream (also what is captured into a graph)."""
        main = self._main
        self._ev_go.record(main)                     # the previous step's packing has read what the chains overwrite
        for _, s in self._sides:
            s.wait_event(self._ev_go)
        def k2():
            if self._k2_batched:     # captures of one length: one transform launch + one finalize for all of them
                self.dev.welch_batch_dev(self.caps, self.nbytes[0], self.chunk_samples, self.nperseg, self.fs, self.psd)
                return
            for a, cap in enumerate(self.caps):
                if self.rows[a]:
                    self.dev.welch_dev(cap, self.nbytes[a], self.chunk_samples, self.nperseg, self.fs, self.psd[a])

        # Which chain is issued first: K2 (default).  With the side chains down to two launches per capture the other
        # order was tried -- the fused scan takes 9 us on a 10-s capture when it has the chip, ~40 us beside a K2 launch --
        # and measured slower, graph 0.300 against 0.262 ms per step, eager 0.273 against 0.270 (two rounds on one box,
        # tools/deployment_probe.py [--k2-first]): the main stream's three K2 launches are still the longer chain, and
        # delaying them costs more than the scans gain.  ``scan_first`` keeps the experiment reachable.
        if not self.scan_first:
            k2()
        for a, cap in enumerate(self.caps):
            sdev, _ = self._sides[a % len(self._sides)]
            # two launches per capture: the fused pass, then the tail (threshold, amplitude totals, onset, slot)
            sdev.capture_scan_dev(cap, self.nbytes[a], self.chunk_bytes, self.power[a], self.rssi_threshold, self.amp[a],
                                  self.noise_samples, self.window, self.factor, self.onset[a],
                                  d_stats=self.stats[a] if self.n_chunks[a] else None, slice_samples=self.slice_samples,
                                  d_slot=self.slots[a])
        if self.scan_first:
            k2()
        sdev0, s0 = self._sides[0]
        for k in range(1, len(self._sides)):         # every slot is in place before the pairs are solved
            self._ev_side[k].record(self._sides[k][1])
            s0.wait_event(self._ev_side[k])
        if self.pairs:
            sdev0.xcorr_slots_dev(self.slots, self.slot_bytes, self.n_ant, self.slice_samples, self.pairs, self.lags,
                                  self.peaks, self.margins)
        self._ev_side[0].record(s0)
        main.wait_event(self._ev_side[0])
        self.dev.pack_results_dev(self._pack_desc(out), self.nperseg, self.d_pairs, self.lags, self.peaks, self.margins)

    def _pack_desc(self, out: torch.Tensor):
        """The captures' descriptors for gj_pack_results_dev (one launch for every antenna's result vector); antenna 0
        carries the pair table."""
        from . import _ffi
        key = out.data_ptr()
        if key not in self._desc:
            d = []
            for a in range(self.n_ant):
                carries = a == 0 and bool(self.pairs)
                d.append(_ffi.CombineCapture(n_chunks=self.n_chunks[a], rows=self.rows[a], n_tiles=0, total_bytes=self.nbytes[a], n_parts=1,
                                             antenna=a, n_pairs=len(self.pairs) if carries else 0, pair_cap=len(self.pairs),
                                             d_power=self.power[a].data_ptr(), d_stats=self.stats[a].data_ptr(), d_tiles=None,
                                             d_amp_parts=None, d_onset_parts=None, d_amp=self.amp[a].data_ptr(),
                                             d_onset=self.onset[a].data_ptr(), d_psd=self.psd[a].data_ptr(), d_out=out[a].data_ptr()))
            self._desc[key] = d
        return self._desc[key]

    def _capture(self, k: int):
        """The step for result set k as ONE HIP graph (stream capture on the main stream; the side streams join the
        capture through the events).
        Beside a live torch.distributed NCCL (= RCCL) process group the capture first lets the group's outstanding
        collectives complete AND its watchdog notice (it polls every 100 ms): the side streams come from torch's stream
        pool like the group's internal stream, and HIP refuses a query of an event whose stream is being captured
        (hipErrorCapturedEvent, raised in the watchdog thread, ends the process).  Collectives issued from ANOTHER thread
        while a step is being captured are not covered: construct with ``graph=False`` then.  Round 4: ~45 launches per step, 0.37-0.38 ms launched one by one against 0.345-0.352 ms
        replayed (the runtime's graph executor starts the branches one after the other: profiles/r04_deployment.txt).
        Round 5: twelve launches per step (one K2 + finalize, two per capture on the side, three for K5, one pack), 0.22 ms
        eager against 0.20-0.21 ms replayed (profiles/r05_deployment_timeline_*.txt).  Everything in a step is capturable --
        kernel launches and event fork / join, no memset, no allocation.  If the runtime refuses the capture the step stays
        eager (same kernels)."""
        try:
            torch.cuda.synchronize()
            try:
                import torch.distributed as dist
                from torch.distributed import distributed_c10d as _c10d
                if dist.is_available() and dist.is_initialized() and any("nccl" in str(v[0]).lower() for v in _c10d._world.pg_map.values()):
                    import time
                    time.sleep(0.15)             # one watchdog period: no collective's end event is left to be queried
            except Exception:                    # noqa: BLE001 -- a torch without distributed support
                pass
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._main, capture_error_mode="thread_local"):
                self._enqueue(self._final[k])
            return g
        except Exception as e:                       # noqa: BLE001 -- any refusal: eager launches from here on
            _log.warning("the step could not be captured into a HIP graph (%r): launching it kernel by kernel", e)
            self._graphs = None
            torch.cuda.synchronize()
            return None

    def step(self) -> StepResults:
        """One pass over every capture + every pair; no host synchronisation.  Read the results through the returned
        StepResults (``unpack()`` / ``wait()``); its buffers are reused two steps later."""
        self._idx ^= 1
        k = self._idx
        done = False
        if self._graphs is not None and self._steps >= 1:          # the first step runs eagerly: lazy one-time set-up
            if self._graphs[k] is None:
                self._graphs[k] = self._capture(k)
            if self._graphs is not None and self._graphs[k] is not None:
                with torch.cuda.stream(self._main):
                    self._graphs[k].replay()
                done = True
        if not done:
            self._enqueue(self._final[k])
        self._steps += 1
        self._done[k].record(self._main)
        return StepResults(self._final[k], self._done[k], self.n_ant)

    def close(self):
        self._graphs = None                              # the captured graphs name this object's buffers
        for sdev, _ in self._sides:
            sdev.close()
        self._sides = []
        for b in getattr(self, "_file_buffers", []):     # from_files: the captures were uploaded for this object
            b.free()
        self._file_buffers = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def from_files(dev, paths: Sequence[str], device: Optional[torch.device] = None, **kw) -> LocalAntennas:
    """The deployment over capture FILES: each file is uploaded once (gj_upload_file) into a tensor-like buffer that
    stays resident; then steps as above."""
    from .split import CaptureRange
    import os
    d = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    bufs = [CaptureRange(dev, p, 0, os.path.getsize(p), d) for p in paths]

    class _View:                                         # what LocalAntennas needs of a capture
        dtype = torch.uint8
        is_cuda = True

        def __init__(self, rng):
            self._r, self.device = rng, d

        def is_contiguous(self):
            return True

        def numel(self):
            return self._r.numel()

        def data_ptr(self):
            return self._r.data_ptr()

    st = LocalAntennas(dev, [_View(b) for b in bufs], **kw)
    st._file_buffers = bufs
    return st
