"""One capture over several GPUs ("fewer files than GPUs => split one file into
contiguous chunk ranges aligned to 65 536 B / 1-s chunks").

The reference's deployment has THREE antennas (GpsJammerApp/app/worker.py:97-101,586-600;
skrypty/triangulateRSSI.py:147-154).  With one capture per GPU an 8-GPU node leaves five GPUs idle and the
per-file latency where it was; here the captures are laid end to end, cut into ``world`` contiguous runs of
*units* (a unit = lcm(power chunk, PSD chunk) = 8 192 000 bytes = 2 s of capture) and every rank works on its
run -- at most one part of two neighbouring captures each.  Per-rank work falls as 1/world: strong scaling.

What a part computes, and why the combined result is bit-identical to the unsplit run (tests pin it):
  K1 chunk powers, K2 PSD rows   per-chunk quantities; parts start on chunk boundaries; the Welch workgroup
                                 split is planned for the whole capture (gj_part_welch_dev)
  K3 amplitude statistics        per-64-KiB-tile sums travel as they are; the combining rank adds them with the
                                 same code in the same order (gj_amp_combine_dev); the first hit and the tail
                                 of its tile come from the part that holds the bytes
  K4 onset                       exact integer window sums; every part brings a halo of one tile in front of
                                 its range and the capture's noise span (each rank reads those 400 KB itself
                                 -- no collective), so the parts' moving-average positions tile the capture;
                                 the smallest index wins (gj_onset_combine_dev)
  TDOA slot                      cut by every part that found an onset in its positions (the buffer has a tail
                                 of one slice behind the own range); gj_slots_pick_dev keeps, per capture, the
                                 slot cut at the smallest onset: the bytes the unsplit capture would have cut
Exchange: ONE all-gather of the parts' slots, then every rank solves its share of the antenna pairs; ONE
gather of the part vectors to rank 0, which rebuilds each capture's arrays in HBM and runs the same tail
kernels (threshold, mean spectrum, packing) as a single-GPU stream -- for ALL captures in three launches
(gj_split_combine_dev: assemble, statistics, pack; the copy list is static and validated once).  ``StepResults`` comes
out as from ``gpsjam.sharded.AntennaStream``: one result vector per ANTENNA.

``emulate=True`` (bench.py --split --emulate-world W): this process is ONE rank of a W-rank plan on a single GPU --
it holds only that rank's parts, the exchange is local copies into the W-rank buffers, and what the other ranks would
have sent is put there beforehand (``adopt_remote``).  Rank 0's step then carries the per-rank load of W GPUs: its
own 1/W of the bytes, its share of the pairs, and the combine over all W ranks' part vectors.
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _ffi
from .sharded import (HEADER, LAG_INVALID, PAIR_FIELDS, StepResults, all_pairs, allgather_rows, gather_rows,
                      result_len, slot_bytes)

TILE = 65536


def unit_bytes(chunk_bytes: int = 65536, chunk_samples: int = 2048000) -> int:
    """Smallest run of capture bytes that is a whole number of power chunks AND of PSD chunks."""
    return math.lcm(int(chunk_bytes), 2 * int(chunk_samples))


@dataclass(frozen=True)
class Part:
    antenna: int
    part: int            # index among the parts of its capture
    parts: int           # parts of its capture
    first_byte: int      # own range [first_byte, first_byte + own_bytes) of the capture
    own_bytes: int
    total_bytes: int     # of the capture
    rank: int
    local: int           # index among the parts of its rank

    @property
    def is_last(self) -> bool:
        return self.first_byte + self.own_bytes == self.total_bytes


def plan_parts(capture_bytes: Sequence[int], world: int, unit: int) -> List[Part]:
    """Lay the captures end to end in units and cut the line into ``world`` contiguous runs of (almost) equal
    length; a run that crosses the boundary between two captures gives its rank one part of each.  The ragged
    end of a capture belongs to its last unit.  Ranks beyond the number of units get nothing.
    (Equal runs on purpose: giving rank 0 -- which also gathers and combines -- a shorter run was measured and buys
    nothing, every rank's step is its K2 plus the same fixed chain of small kernels: profiles/r04_split_emulated.txt.)"""
    units = [max(1, -(-int(b) // unit)) for b in capture_bytes]
    total = sum(units)
    bounds = [(r * total) // world for r in range(world + 1)]
    start_of = [sum(units[:a]) for a in range(len(units))]
    raw = []
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        for a, (s0, n) in enumerate(zip(start_of, units)):
            u0, u1 = max(lo, s0), min(hi, s0 + n)
            if u0 < u1:
                first = (u0 - s0) * unit
                end = min((u1 - s0) * unit, int(capture_bytes[a])) if u1 - s0 < n else int(capture_bytes[a])
                raw.append((a, first, end - first, r))
    out, per_rank = [], {}
    for a in range(len(units)):
        mine = [x for x in raw if x[0] == a]
        for g, (_, first, own, r) in enumerate(mine):
            out.append((a, g, len(mine), first, own, int(capture_bytes[a]), r))
    out.sort(key=lambda x: (x[6], x[0], x[1]))
    parts = []
    for a, g, G, first, own, tot, r in out:
        j = per_rank.get(r, 0)
        per_rank[r] = j + 1
        parts.append(Part(a, g, G, first, own, tot, r, j))
    return parts


def halo_bytes(part: Part, window: int) -> int:
    """Whole tiles in front of the own range holding at least window - 1 samples (none for a capture's first part)."""
    if part.first_byte == 0:
        return 0
    return -(-2 * (window - 1) // TILE) * TILE


def buffer_range(part: Part, window: int, slice_samples: int) -> Tuple[int, int]:
    """Capture bytes [b0, b1) a rank keeps in HBM for a part: halo + own range + one TDOA slice of tail."""
    b0 = part.first_byte - halo_bytes(part, window)
    tail = -(-2 * slice_samples // TILE) * TILE
    return b0, min(part.total_bytes, part.first_byte + part.own_bytes + tail)


def deal_pairs(n_ant: int, ranks_with_parts: Sequence[int]) -> dict:
    """Every antenna pair once, dealt round-robin over the ranks that hold a part."""
    deal = {r: [] for r in ranks_with_parts}
    order = sorted(ranks_with_parts)
    for k, p in enumerate(all_pairs(n_ant)):
        deal[order[k % len(order)]].append(p)
    return deal


class CaptureRange:
    """Bytes [b0, b1) of a capture FILE in HBM (gj_upload_file with an offset: every rank reads its own range, no
    rank ever reads a whole file), with the three attributes PartStream needs of a buffer."""

    dtype = torch.uint8

    def __init__(self, dev, path, b0: int, b1: int, device):
        self._cap = dev.capture(path, offset=int(b0), max_bytes=int(b1 - b0)) if b1 > b0 else None
        self._n = int(b1 - b0)
        self.device = device
        if self._cap is not None and self._cap.nbytes != self._n:
            raise ValueError(f"{path}: wanted bytes [{b0}, {b1}), the file gave {self._cap.nbytes}")

    def data_ptr(self) -> int:
        return self._cap.ptr if self._cap is not None else 0

    def numel(self) -> int:
        return self._n

    def is_contiguous(self) -> bool:
        return True

    def free(self):
        if self._cap is not None:
            self._cap.free()
            self._cap = None


def from_files(dev, paths: Sequence[str], *, rank: int = 0, world_size: int = 1, device=None, **kw) -> "SplitStreams":
    """The split pipeline over capture FILES (the reference's deployment: three antenna recordings,
    GpsJammerApp/app/worker.py:586-600): the files are laid end to end and cut into ``world_size`` runs; this rank
    uploads only the byte ranges of its own parts (+ halo, tail and, for parts that do not start their file, the first
    400 000 bytes for K4's threshold)."""
    import os
    sizes = [os.path.getsize(p) for p in paths]
    d = device if device is not None else torch.device("cuda", torch.cuda.current_device())

    def make_buffer(part, b0, b1):
        return CaptureRange(dev, paths[part.antenna], b0, b1, d)

    def make_noise(antenna, nbytes):
        return CaptureRange(dev, paths[antenna], 0, nbytes, d)

    return SplitStreams(dev, sizes, make_buffer, make_noise, rank=rank, world_size=world_size, device=d, **kw)


def emulated_rank(dev, capture_bytes: Sequence[int], make_buffer, make_noise, world: int, rank: int = 0, **kw) -> "SplitStreams":
    """Rank ``rank`` of a ``world``-rank split run, alone on ONE GPU, with what the other ranks would have sent already in
    place (bench.py --split --emulate-world W [--emulate-rank R]; tests): every other rank of the plan is walked once on
    this GPU -- its parts scanned, transformed, its slots cut; then, over everybody's slots, its share of the pairs
    solved and its part vectors packed -- and its slots (and, for rank 0, its vectors) are handed to the timed rank
    (``adopt_remote``).  Its steps afterwards do what that rank of ``world`` GPUs does: 1/world of the bytes, its pairs,
    and on rank 0 the combine over ALL ranks' vectors."""
    other_kw = dict(kw, overlap=False, exchange_always=False)
    me = SplitStreams(dev, capture_bytes, make_buffer, make_noise, rank=rank, world_size=world, emulate=True, **kw)
    others = [SplitStreams(dev, capture_bytes, make_buffer, make_noise, rank=r, world_size=world, emulate=True, **other_kw)
              for r in range(world) if r != rank]
    for st in [me] + others:                              # pass 1: what needs no other rank
        st.scan()
        with st._on_side():
            st.cut_slots()
    torch.cuda.synchronize()
    for st in others:
        me.all_slots[st.rank * st.pmax:(st.rank + 1) * st.pmax].copy_(st.my_slots)
    me.all_slots[me.rank * me.pmax:(me.rank + 1) * me.pmax].copy_(me.my_slots)
    for st in others:                                     # pass 2: over everybody's slots
        st.all_slots.copy_(me.all_slots)
        st.solve(st.all_slots)
        vec = st.pack()
        torch.cuda.synchronize()
        me.adopt_remote(st.rank, st.my_slots, vec)
    torch.cuda.synchronize()
    for st in others:
        st.close()
    if me._main is not None:                              # the walked ranks pointed the shared context at their streams
        dev.set_stream(me._main.cuda_stream)
    return me


def emulated_rank0(dev, capture_bytes: Sequence[int], make_buffer, make_noise, world: int, **kw) -> "SplitStreams":
    return emulated_rank(dev, capture_bytes, make_buffer, make_noise, world, 0, **kw)


class PartStream:
    """One part on this GPU: its buffers and the three per-part kernels (no host synchronisation anywhere)."""

    def __init__(self, dev, dev_side, part: Part, buf: torch.Tensor, noise: Optional[torch.Tensor], *, chunk_bytes,
                 chunk_samples, nperseg, fs, slice_samples, noise_samples, window, factor, rssi_threshold):
        b0, b1 = buffer_range(part, window, slice_samples)
        assert buf.dtype == torch.uint8 and buf.is_contiguous() and buf.numel() == b1 - b0, (buf.numel(), b1 - b0)
        if part.first_byte or part.own_bytes < 2 * noise_samples:
            assert noise is not None and noise.numel() >= min(2 * noise_samples, part.total_bytes)
        self.part, self.dev, self.dev_side, self.buf, self.noise = part, dev, dev_side, buf, noise
        self.chunk_bytes, self.chunk_samples, self.nperseg, self.fs = chunk_bytes, chunk_samples, nperseg, fs
        self.slice_samples, self.noise_samples, self.window, self.factor = slice_samples, noise_samples, window, factor
        self.rssi_threshold = rssi_threshold
        self.view = _ffi.PartView(buf.data_ptr(), buf.numel(), b0, part.first_byte, part.own_bytes, part.total_bytes,
                                  noise.data_ptr() if noise is not None else None)
        d = buf.device
        self.n_chunks = dev.chunk_count(part.own_bytes, chunk_bytes)
        self.rows = dev.welch_rows(part.own_bytes, chunk_samples, nperseg)
        self.n_tiles = dev.amp_tile_count(part.own_bytes)
        self.first_chunk = part.first_byte // chunk_bytes
        self.first_row = part.first_byte // (2 * chunk_samples)
        self.first_tile = part.first_byte // TILE
        self.power = torch.empty(max(self.n_chunks, 1), dtype=torch.float32, device=d)
        self.tiles = torch.zeros(2 * max(self.n_tiles, 1), dtype=torch.float64, device=d)   # (sum, first) records
        self.amp = torch.zeros(4, dtype=torch.int64, device=d)                                 # gj_amp_part
        self.onset = torch.zeros(4, dtype=torch.int64, device=d)                               # gj_onset
        # two PSD buffers, written alternately: a step's packing (which reads one) may run on the second stream while the
        # main stream has gone on to the next step's K2 + finalize (which writes the other)
        self.psd2 = [torch.empty((max(self.rows, 1), nperseg), dtype=torch.float32, device=d) for _ in range(2)]
        self.psd = self.psd2[0]

    def scan(self, slot: Optional[torch.Tensor] = None):
        """K1 + K3 + K4 of the part in two launches; with ``slot`` the TDOA slot at the part's own onset is cut by the
        same tail launch (gj_part_capture_scan_dev) and the next ``slot(out)`` for that buffer has nothing left to do."""
        self._slot_cut = None
        if slot is None:
            self.dev_side.part_scan_dev(self.view, self.chunk_bytes, self.power, self.rssi_threshold, self.tiles, self.amp,
                                        self.noise_samples, self.window, self.factor, self.onset)
        else:
            self.dev_side.part_capture_scan_dev(self.view, self.chunk_bytes, self.power, self.rssi_threshold, self.tiles,
                                                self.amp, self.noise_samples, self.window, self.factor, self.onset,
                                                slice_samples=self.slice_samples, d_slot=slot)
            self._slot_cut = slot.data_ptr()

    def welch(self, which: int = 0):
        self.psd = self.psd2[which]
        self.dev.part_welch_dev(self.view, self.chunk_samples, self.nperseg, self.fs, self.psd)

    def slot(self, out: torch.Tensor):
        if getattr(self, "_slot_cut", None) == out.data_ptr():     # cut by this step's scan already
            self._slot_cut = None
            return
        self.dev_side.part_slot_dev(self.view, self.onset, self.slice_samples, out)


class SplitStreams:
    """This rank's share of ``capture_bytes`` (one entry per antenna) cut by ``plan_parts``.

    ``make_buffer(part, b0, b1)`` returns the uint8 device tensor holding capture bytes [b0, b1) of the part's
    antenna; ``make_noise(antenna, nbytes)`` the capture's first ``nbytes`` bytes (asked for every part that does
    not start its capture).  Both are called at construction, for this rank's parts only."""

    def __init__(self, dev, capture_bytes: Sequence[int], make_buffer, make_noise, *, rank: int = 0, world_size: int = 1,
                 chunk_bytes: int = 65536, chunk_samples: int = 2048000, nperseg: int = 4096, fs: float = 2.048e6,
                 slice_samples: int = 1 << 19, noise_samples: int = 200000, window: int = 1000, factor: float = 50.0,
                 rssi_threshold: float = 0.0, overlap: Optional[bool] = None, device=None, exchange_always: bool = False,
                 emulate: bool = False, pack_on_side: bool = True, group=None):
        self.dev, self.rank, self.world = dev, rank, world_size
        self.group = group                 # torch.distributed group of the exchange (None: the default group)
        # one rank of a world_size-rank plan alone on its GPU: no collective, local copies into the world-size buffers
        self.emulate = bool(emulate)
        # a process group of one: still issue the slot all-gather and the part gather (the collective path on one GPU)
        self._always = bool(exchange_always) and (world_size == 1 or self.emulate)
        self.capture_bytes = [int(b) for b in capture_bytes]
        self.n_ant = len(self.capture_bytes)
        self.chunk_bytes, self.chunk_samples, self.nperseg, self.fs = chunk_bytes, chunk_samples, nperseg, fs
        self.slice_samples, self.noise_samples, self.window, self.factor = slice_samples, noise_samples, window, factor
        self.unit = unit_bytes(chunk_bytes, chunk_samples)
        self.parts = plan_parts(self.capture_bytes, world_size, self.unit)
        self.mine = [p for p in self.parts if p.rank == rank]
        self.pmax = max(1, max((p.local for p in self.parts), default=0) + 1)
        d = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = d
        self.is_root = rank == 0
        self.overlap = bool(d.type == "cuda" if overlap is None else overlap)
        self.dev_side = dev
        self.streams_overlap = None        # one stream: nothing to overlap
        self._pack_on_side = bool(pack_on_side)
        self._pidx = 0
        self._main = torch.cuda.current_stream(d) if d.type == "cuda" else None
        if self._main is not None:
            dev.set_stream(self._main.cuda_stream)
        if self.overlap:
            from .streams import stream_beside_checked
            self.dev_side = type(dev)(dev.index)
            # on a hardware queue of its own (gpsjam/streams.py); streams_overlap False: none found, chains serialised
            self._side, self.streams_overlap = stream_beside_checked([(dev, self._main)], device=d)
            self.dev_side.set_stream(self._side.cuda_stream)
            self._ev_free, self._ev_side, self._ev_packed = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            self._ev_free.record(self._main)
            # Packing on the SECOND stream (behind K5, where its inputs come from): the main stream then carries nothing but
            # K2 + finalize, step after step, and never waits for the side chain -- at the per-rank load of an eight-way
            # split the wait, the packing launch and the two cross-stream joins were 45 us of a 0.58-ms step between two K2
            # launches (profiles/NOTES_r05.md).  Needs the two PSD buffers of PartStream.
            self._ev_psd = [torch.cuda.Event(), torch.cuda.Event()]         # main: finalize has written psd2[i]
            self._ev_psd_read = [None, None]                                # side: the packing has read psd2[i]
        else:
            self._side = self._main
        # Rank 0's gather + combine get a THIRD stream (and a context bound to it): on the second stream they would sit
        # between step k's K5 and step k + 1's scan, and under K2 -- where every small launch waits tens of microseconds
        # for a slot -- that serialisation left the next scan starting when K2 was two thirds through
        # (profiles/r04_split_emulated8_timeline.txt).  The combine reads only the packed vectors of its own step.
        self.dev_comb, self._comb = self.dev_side, self._side
        if self.overlap and self.is_root:
            self.dev_comb = type(dev)(dev.index)
            # The runtime deals streams over a few hardware queues and two streams on one queue run one after the other: a
            # third stream once landed on the MAIN stream's queue and K2 queued behind the combine.  stream_beside tests
            # candidates until one runs beside both other streams; more queues (GPU_MAX_HW_QUEUES=8) give it room.
            self._comb, ok = stream_beside_checked([(dev, self._main), (self.dev_side, self._side)], device=d)
            self.streams_overlap = self.streams_overlap and ok
            self.dev_comb.set_stream(self._comb.cuda_stream)
        kw = dict(chunk_bytes=chunk_bytes, chunk_samples=chunk_samples, nperseg=nperseg, fs=fs, slice_samples=slice_samples,
                  noise_samples=noise_samples, window=window, factor=factor, rssi_threshold=rssi_threshold)
        self.streams: List[PartStream] = []
        for p in self.mine:
            b0, b1 = buffer_range(p, window, slice_samples)
            # K4's threshold needs the capture's first noise_samples samples: every part that does not hold them
            # itself brings its own copy (a few hundred KB read by each rank -- no collective)
            need_noise = p.first_byte or p.own_bytes < 2 * noise_samples
            noise = make_noise(p.antenna, min(2 * noise_samples, p.total_bytes)) if need_noise else None
            self.streams.append(PartStream(dev, self.dev_side, p, make_buffer(p, b0, b1), noise, **kw))
        # capacities of a part vector: the largest part of the plan (every rank knows the whole plan)
        self.chunk_cap = max(dev.chunk_count(p.own_bytes, chunk_bytes) for p in self.parts)
        self.tile_cap = max(dev.amp_tile_count(p.own_bytes) for p in self.parts)
        self.rows_cap = max(max(dev.welch_rows(p.own_bytes, chunk_samples, nperseg) for p in self.parts), 1)
        ranks_with_parts = sorted({p.rank for p in self.parts})
        self.deal = deal_pairs(self.n_ant, ranks_with_parts)
        self.pairs = self.deal.get(rank, [])
        self.pair_cap = max(1, max(len(v) for v in self.deal.values()))
        all_solved = [x for r in sorted(self.deal) for p in self.deal[r] for x in p]   # rank order = the combine's order
        self._d_all_pairs = torch.tensor(all_solved or [0, 0], dtype=torch.int32, device=d)
        self.part_len = dev.part_result_len(self.chunk_cap, self.tile_cap, self.rows_cap, nperseg, self.pair_cap)
        self.o_tiles = HEADER + self.chunk_cap
        self.o_pairs = self.o_tiles + 2 * self.tile_cap
        self.o_rows = self.o_pairs + PAIR_FIELDS * self.pair_cap
        # slots: this rank's (pmax of them, unused ones stay invalid), everybody's, one per antenna
        self.slot_bytes = slot_bytes(slice_samples)
        self.my_slots = torch.zeros((self.pmax, self.slot_bytes), dtype=torch.uint8, device=d)
        self._invalidate(self.my_slots)
        self.all_slots = torch.zeros((world_size * self.pmax, self.slot_bytes), dtype=torch.uint8, device=d)
        self.ant_slots = torch.zeros((self.n_ant, self.slot_bytes), dtype=torch.uint8, device=d)
        members, offsets = [], [0]
        for a in range(self.n_ant):
            members += [p.rank * self.pmax + p.local for p in self.parts if p.antenna == a]
            offsets.append(len(members))
        self.d_members = torch.tensor(members or [0], dtype=torch.int32, device=d)
        self.d_offsets = torch.tensor(offsets, dtype=torch.int32, device=d)
        npairs = max(len(self.pairs), 1)
        self.d_pairs = torch.tensor([x for p in self.pairs for x in p] or [0, 0], dtype=torch.int32, device=d)
        self.lags = torch.full((npairs,), LAG_INVALID, dtype=torch.int32, device=d)
        self.peaks = torch.zeros(npairs, dtype=torch.float32, device=d)
        self.margins = torch.zeros(npairs, dtype=torch.float32, device=d)
        # vectors: this rank's parts (two sets, used alternately), everybody's on the root, one per antenna on the root
        self._vecs = [torch.zeros((self.pmax, self.part_len), dtype=torch.float64, device=d) for _ in range(2)]
        self._gathered = ([torch.zeros((world_size, self.pmax * self.part_len), dtype=torch.float64, device=d) for _ in range(2)]
                          if self.is_root else [None, None])
        self.n_chunks_of = [dev.chunk_count(b, chunk_bytes) for b in self.capture_bytes]
        self.rows_of = [dev.welch_rows(b, chunk_samples, nperseg) for b in self.capture_bytes]
        self.total_pairs = self.n_ant * (self.n_ant - 1) // 2
        self.final_len = max(result_len(n, nperseg, self.total_pairs) for n in self.n_chunks_of)
        # rank 0: two arenas (used alternately, like the vectors) with every capture's assembled arrays and result
        # vectors, and the static plan of the three-launch combine over each
        self._final, self._psd_views, self._plans, self._arenas = [None, None], [None, None], [None, None], [None, None]
        if self.is_root:
            for k in range(2):
                self._build_combine(k)
        if self.emulate and self._always:      # the collectives of a one-rank group, issued beside the local copies
            self._one_slots = torch.zeros((1, self.pmax * self.slot_bytes), dtype=torch.uint8, device=d)
            self._one_vec = torch.zeros((1, self.pmax * self.part_len), dtype=torch.float64, device=d)
        self._done = [torch.cuda.Event() for _ in range(2)] if d.type == "cuda" else [None, None]
        self._ev_vec_free = [None, None]
        self._idx = 0
        self.last_psd = [None] * self.n_ant     # rank 0: each capture's waterfall rows as rebuilt by the last combine
        # workspaces
        ws_main = max([dev.part_welch_workspace(s.view, chunk_samples, nperseg) for s in self.streams] + [1 << 20])
        ants = len({a for p in self.pairs for a in p}) or 1
        ws_side = max(dev.xcorr_workspace(ants, slice_samples, npairs),
                      max([(s.buf.numel()) // 48 + (1 << 20) for s in self.streams] + [1 << 20]))
        if self.overlap:
            dev.reserve(ws_main)
            self.dev_side.reserve(ws_side)
        else:
            dev.reserve(max(ws_main, ws_side))

    @staticmethod
    def _invalidate(slots: torch.Tensor):
        """Slot headers -> (flag -1, start -1): an unused slot must never be picked."""
        slots.view(torch.int64)[:, :2] = -1

    def _on_side(self):
        return torch.cuda.stream(self._side) if self.overlap else contextlib.nullcontext()

    def _on_comb(self):
        return torch.cuda.stream(self._comb) if self.overlap else contextlib.nullcontext()

    # ---------------------------------------------------------------- the step
    def stream_scan(self):
        """K1 + K3 + K4 of every part of this rank, one fused pass each (side stream)."""
        if self.overlap:
            self._side.wait_event(self._ev_free)
        for j, s in enumerate(self.streams):
            s.scan(slot=self.my_slots[j])
        if self.overlap:
            self._ev_side.record(self._side)

    def welch(self):
        """K2 of every part of this rank (main stream)."""
        side_pack = self.overlap and self._pack_on_side
        if side_pack:
            self._pidx ^= 1
            if self._ev_psd_read[self._pidx] is not None:           # the packing of two steps ago has read this buffer
                self._main.wait_event(self._ev_psd_read[self._pidx])
        for s in self.streams:
            s.welch(self._pidx)
        if side_pack:
            self._ev_psd[self._pidx].record(self._main)

    def scan(self):
        self.stream_scan()
        self.welch()

    def cut_slots(self):
        """Slots of this rank's parts (current stream)."""
        for j, s in enumerate(self.streams):
            s.slot(self.my_slots[j])

    def solve(self, slots: torch.Tensor):
        """One slot per antenna out of everybody's, then this rank's share of the pairs (current stream)."""
        self.dev_side.slots_pick_dev(slots, self.slot_bytes, self.d_offsets, self.d_members, self.n_ant, self.ant_slots)
        if self.pairs:
            self.dev_side.xcorr_slots_dev(self.ant_slots, self.slot_bytes, self.n_ant, self.slice_samples, self.pairs,
                                          self.lags, self.peaks, self.margins)

    def tdoa(self):
        """Slots of this rank's parts -> ONE all-gather -> one slot per antenna -> this rank's pairs."""
        with self._on_side():
            self.cut_slots()
            if self.emulate:
                # alone on the GPU: this rank's rows of the world-size buffer are filled by a copy (the other ranks'
                # rows were put there by adopt_remote); with exchange_always the all-gather of a one-rank group is
                # issued as well, so that the RCCL call sits in the chain where N ranks have it
                src = self.my_slots
                if self._always:
                    src = allgather_rows(self.my_slots.view(-1), 1, out=self._one_slots, always=True, group=self.group).view(self.pmax, -1)
                self.all_slots[self.rank * self.pmax:(self.rank + 1) * self.pmax].copy_(src)
                slots = self.all_slots
            elif self.world > 1 or self._always:
                allgather_rows(self.my_slots.view(-1), self.world, out=self.all_slots.view(self.world, -1), always=self._always, group=self.group)
                slots = self.all_slots
            else:
                slots = self.my_slots
            self.solve(slots)
            if self.overlap:
                self._ev_side.record(self._side)

    def pack(self) -> torch.Tensor:
        side_pack = self.overlap and self._pack_on_side
        stream = self._side if side_pack else self._main
        dev = self.dev_side if side_pack else self.dev
        if side_pack:
            stream.wait_event(self._ev_psd[self._pidx])             # this step's PSD rows (main stream) are written
        elif self.overlap:
            self._main.wait_event(self._ev_side)
        self._idx ^= 1
        vec = self._vecs[self._idx]
        if self.overlap and self._ev_vec_free[self._idx] is not None:
            stream.wait_event(self._ev_vec_free[self._idx])         # the gather / combine of two steps ago has read it
        for j, s in enumerate(self.streams):
            p = s.part
            carries = j == 0                      # the pairs this rank solved ride on its first part's vector
            a = _ffi.PartPack(self.rank, p.antenna, p.part, p.parts, s.first_chunk, s.n_chunks, self.chunk_cap,
                              s.first_row, s.rows, self.rows_cap, s.first_tile, s.n_tiles, self.tile_cap,
                              p.first_byte // 2, self.nperseg, len(self.pairs) if carries else 0, self.pair_cap, 0,
                              s.power.data_ptr(), s.amp.data_ptr(), s.onset.data_ptr(), s.tiles.data_ptr(),
                              s.psd.data_ptr(), self.d_pairs.data_ptr(), self.lags.data_ptr(), self.peaks.data_ptr(),
                              self.margins.data_ptr())
            dev.pack_part_dev(a, vec[j])
        if self.overlap:
            self._ev_free.record(stream)
            self._ev_packed.record(stream)
            if side_pack:
                if self._ev_psd_read[self._pidx] is None:
                    self._ev_psd_read[self._pidx] = torch.cuda.Event()
                self._ev_psd_read[self._pidx].record(stream)
        return vec

    def exchange(self, dst: int = 0) -> Optional[StepResults]:
        assert dst == 0
        # Gather and (rank 0) combine run beside the main stream: a chain of small latency-bound kernels that would
        # otherwise sit between two steps' K2 launches with the chip idle; the main stream goes straight on to the next
        # step.  On rank 0 the chain has a stream of its own (see __init__), elsewhere it follows K5 on the second one.
        # Everything it reads is in the packed vectors (two sets, used alternately).
        vec = self.pack()
        k = self._idx
        final = None
        if self.overlap:
            self._comb.wait_event(self._ev_packed)
        with self._on_comb():
            if self.emulate:
                src = vec.view(-1)
                if self._always:
                    src = gather_rows(vec.view(-1), 0, 1, 0, out=self._one_vec, always=True, group=self.group)[0]
                rows = self._gathered[k]
                if rows is not None:
                    rows[self.rank].copy_(src)
            elif self.world > 1 or self._always:
                rows = gather_rows(vec.view(-1), self.rank, self.world, 0, out=self._gathered[k], always=self._always, group=self.group)
            else:
                rows = vec.view(1, -1)
            if self.is_root:
                final = self._combine(rows, k)
                if self._done[k] is not None:
                    self._done[k].record(self._comb)
            if self.overlap:
                if self._ev_vec_free[k] is None:
                    self._ev_vec_free[k] = torch.cuda.Event()
                self._ev_vec_free[k].record(self._comb)
        if not self.is_root:
            return None
        return StepResults(final, self._done[k], self.n_ant)

    def step(self) -> Optional[StepResults]:
        self.scan()
        self.tdoa()
        return self.exchange(0)

    # ---------------------------------------------------------------- rank 0: parts -> captures
    def _build_combine(self, k: int):
        """Arena k (every capture's assembled arrays + the result vectors) and the static plan of its combine: the copy
        list part vector -> capture-order array and one descriptor per capture, checked on the host and uploaded once
        (gj_combine_plan_create).  Which rank holds which part, and which rank solved which pair, never changes."""
        dev, nper, L8 = self.dev_comb, self.nperseg, 8 * self.part_len
        off = 0

        def take(nbytes: int) -> int:
            nonlocal off
            o = (off + 255) // 256 * 256
            off = o + max(int(nbytes), 8)
            return o

        lay = []
        for a in range(self.n_ant):
            mine = [p for p in self.parts if p.antenna == a]
            lay.append(dict(parts=mine, n_tiles=dev.amp_tile_count(self.capture_bytes[a]),
                            power=take(4 * self.n_chunks_of[a]), stats=take(12),
                            tiles=take(16 * dev.amp_tile_count(self.capture_bytes[a])), amp_parts=take(32 * len(mine)),
                            onset_parts=take(32 * len(mine)), amp=take(32), onset=take(32),
                            psd=take(4 * max(self.rows_of[a], 1) * nper)))
        o_final = take(8 * self.n_ant * self.final_len)
        solved = [(r, self.deal[r]) for r in sorted(self.deal) if self.deal[r]]     # rank order = _d_all_pairs' order
        n_solved = sum(len(v) for _, v in solved)
        o_lags, o_peaks, o_margs = take(4 * max(n_solved, 1)), take(4 * max(n_solved, 1)), take(4 * max(n_solved, 1))
        arena = torch.zeros(off, dtype=torch.uint8, device=self.device)
        base = arena.data_ptr()
        copies, caps = [], []
        for a, ly in enumerate(lay):
            c_off = t_off = r_off = 0
            for g, p in enumerate(ly["parts"]):
                src = (p.rank * self.pmax + p.local) * L8                         # this part's vector in the gathered rows
                nc, nt = dev.chunk_count(p.own_bytes, self.chunk_bytes), dev.amp_tile_count(p.own_bytes)
                nr = dev.welch_rows(p.own_bytes, self.chunk_samples, nper)
                copies.append(_ffi.CombineCopy(src + 8 * HEADER, base + ly["power"] + 4 * c_off, nc, 8, _ffi.GJ_COPY_F64_F32))
                copies.append(_ffi.CombineCopy(src + 8 * self.o_tiles, base + ly["tiles"] + 16 * t_off, 2 * nt, 8, _ffi.GJ_COPY_F64))
                copies.append(_ffi.CombineCopy(src + 8 * 32, base + ly["onset_parts"] + 32 * g, 4, 8, _ffi.GJ_COPY_F64))
                copies.append(_ffi.CombineCopy(src + 8 * 36, base + ly["amp_parts"] + 32 * g, 4, 8, _ffi.GJ_COPY_F64))
                if nr:
                    copies.append(_ffi.CombineCopy(src + 8 * self.o_rows, base + ly["psd"] + 4 * r_off * nper, nr * nper, 4,
                                                   _ffi.GJ_COPY_F32))
                c_off, t_off, r_off = c_off + nc, t_off + nt, r_off + nr
            assert c_off == self.n_chunks_of[a] and t_off == ly["n_tiles"] and r_off == self.rows_of[a], (a, c_off, t_off, r_off)
            carries = a == 0 and n_solved > 0
            caps.append(_ffi.CombineCapture(self.n_chunks_of[a], self.rows_of[a], ly["n_tiles"], self.capture_bytes[a],
                                            len(ly["parts"]), a, n_solved if carries else 0, self.total_pairs,
                                            base + ly["power"], base + ly["stats"], base + ly["tiles"], base + ly["amp_parts"],
                                            base + ly["onset_parts"], base + ly["amp"], base + ly["onset"], base + ly["psd"],
                                            base + o_final + 8 * a * self.final_len))
        q = 0
        for r, prs in solved:                    # the pairs a rank solved ride on its FIRST part's vector
            src = (r * self.pmax) * L8 + 8 * self.o_pairs
            stride = 8 * PAIR_FIELDS
            copies.append(_ffi.CombineCopy(src + 8 * 2, base + o_lags + 4 * q, len(prs), stride, _ffi.GJ_COPY_F64_I32))
            copies.append(_ffi.CombineCopy(src + 8 * 3, base + o_peaks + 4 * q, len(prs), stride, _ffi.GJ_COPY_F64_F32))
            copies.append(_ffi.CombineCopy(src + 8 * 4, base + o_margs + 4 * q, len(prs), stride, _ffi.GJ_COPY_F64_F32))
            q += len(prs)
        self._arenas[k] = arena
        self._final[k] = arena[o_final:o_final + 8 * self.n_ant * self.final_len].view(torch.float64).view(self.n_ant, self.final_len)
        self._psd_views[k] = [arena[ly["psd"]:ly["psd"] + 4 * max(self.rows_of[a], 1) * nper].view(torch.float32)
                              .view(max(self.rows_of[a], 1), nper)[:self.rows_of[a]] for a, ly in enumerate(lay)]
        rows_bytes = self.world * self.pmax * L8
        self._plans[k] = dev.combine_plan(copies, caps, rows_bytes, arena, nper, self._d_all_pairs, base + o_lags,
                                          base + o_peaks, base + o_margs)
        self.combine_launches = 3                # assemble, statistics, pack -- whatever the number of antennas

    def _combine(self, rows: torch.Tensor, k: int) -> torch.Tensor:
        """Every capture rebuilt from its parts' vectors and finished (threshold, amplitude totals, onset, mean
        spectrum, packing) in three launches on the second stream, no host synchronisation, no allocation."""
        assert rows.is_contiguous() and rows.numel() * 8 == self.world * self.pmax * 8 * self.part_len
        self.dev_comb.split_combine_dev(self._plans[k], rows)
        self.last_psd = self._psd_views[k]
        return self._final[k]

    def adopt_remote(self, rank: int, slots: torch.Tensor, vec: torch.Tensor):
        """``emulate`` only: what rank ``rank`` of the plan would have contributed -- its parts' slots ([pmax,
        slot_bytes]) and its packed part vectors ([pmax, part_len]) -- placed where the collectives would have put
        them, once, before the steps."""
        assert self.emulate and rank != self.rank and 0 <= rank < self.world
        self.all_slots[rank * self.pmax:(rank + 1) * self.pmax].copy_(slots)
        for g in self._gathered:
            if g is not None:
                g[rank].copy_(vec.reshape(-1))

    def close(self):
        for k, pl in enumerate(self._plans):
            if pl is not None:
                self.dev_comb.combine_plan_destroy(pl)
                self._plans[k] = None
        if self.dev_comb is not self.dev_side and self.dev_comb is not self.dev:
            self.dev_comb.close()
        if self.overlap and self.dev_side is not self.dev:
            self.dev_side.close()
