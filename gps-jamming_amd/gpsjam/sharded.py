"""One antenna capture per GPU (one process per GPU; RCCL over xGMI through torch.distributed
or through the library's own gj_comm_* entry points).

The per-capture kernels (K1 power scan, K2 Welch PSD, K3 amplitude statistics, K4 onset) are
independent across captures, so captures are sharded one per rank with no data-path
collective.  The one real exchange of the path is TDOA (skrypty/triangulateTDOA.py:60-90
generalised from two antennas to N): every rank cuts its onset-aligned slice into a *TDOA
slot* (16-byte validity header + 2 bytes per sample: 1 MiB for 2^19 samples) and ONE all-gather
puts all N slots on every rank.  EVERY antenna pair (i, j) is solved, and the N(N-1)/2 pairs are
dealt over the ranks so that the per-rank work stays constant as N grows: rank r solves
(r, r+d mod N) for d = 1 .. N/2 (``pairs_of_rank``: 3 pairs on one rank for N = 1 + 2 auxiliary
slots = BASELINE configs[3]; 3-4 pairs per rank over 5 antennas for N = 8) with one multi-pair K5
launch that transforms only the slots its pairs name.  One gather then brings each rank's
fixed-layout result vector -- power map, noise floor, threshold, amplitude statistics, onset, mean
spectrum and the {i, j, lag, peak, margin} of the pairs it solved -- to rank 0, which runs the
host-side solvers (grid search / bearing).  An un-found onset or a slice that runs off the end
of a capture marks that slot invalid and every pair with that antenna comes back as
GJ_LAG_INVALID (the reference aborts there, triangulateTDOA.py:67-77).

``pack_results`` / ``unpack_results`` / ``make_slot`` / ``gather_rows`` / ``allgather_rows`` only
touch torch tensors and torch.distributed, so they run unchanged on CPU tensors with the gloo
backend (tests) and on HIP tensors with the nccl (= RCCL) backend (bench.py).
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple  # noqa: F401

import logging

import numpy as np
import torch

_log = logging.getLogger("gpsjam.sharded")

HEADER = 40          # GJ_RESULT_HEADER: scalars in front of the vectors, see RESULT_FIELDS
RESULT_FIELDS = ("n_chunks", "baseline", "threshold", "n_above", "amp_first", "amp_count",
                 "amp_mean", "onset", "lag", "peak", "noise_power", "n_rows", "nperseg",
                 "rank", "n_pairs", "pair_capacity", "onset_margin_hit", "onset_margin_before",
                 "onset_guard", "onset_threshold", "antenna", "part", "parts", "first_chunk",
                 "first_row", "first_sample", "amp_sum", "amp_tail", "tiles", "first_tile",
                 "reserved0", "reserved1",
                 "onset_record0", "onset_record1", "onset_record2", "onset_record3",      # gj_onset as it is (32 bytes)
                 "amp_record0", "amp_record1", "amp_record2", "amp_record3")              # gj_amp_stats / gj_amp_part
ONSET_NEAR_TIE = 1e-6     # gj_onset: the rounding band of K4's decision (include/gpsjam.h)
LAG_NEAR_TIE = 2e-5       # K5: 1 - runner-up/peak below the rounding of a complex64 FFT
SLOT_HEADER = 16     # GJ_SLOT_HEADER: int64 flag (0 valid / -1 invalid), int64 start sample
LAG_INVALID = -(1 << 31)


PAIR_FIELDS = 5      # GJ_RESULT_PAIR_FIELDS: i, j, lag, peak, margin


def result_len(n_chunks: int, nperseg: int, pair_capacity: int = 0) -> int:
    return HEADER + n_chunks + nperseg + PAIR_FIELDS * pair_capacity


def slot_bytes(n_samples: int) -> int:
    """gj_tdoa_slot_bytes: header + slice, padded to a multiple of 256 bytes."""
    return (SLOT_HEADER + 2 * n_samples + 255) // 256 * 256


def all_pairs(n_ant: int) -> List[Tuple[int, int]]:
    """(i, j), i < j: lag of antenna j relative to antenna i, for every pair
    (BASELINE configs[3]: 3 antennas -> (0,1), (0,2), (1,2))."""
    return [(i, j) for i in range(n_ant) for j in range(i + 1, n_ant)]


def pairs_of_rank(rank: int, world: int) -> List[Tuple[int, int]]:
    """The pairs rank ``rank`` of ``world`` solves: (rank, rank + d mod world) for d = 1 .. (world-1)//2, and for
    even ``world`` the diameter d = world/2 on the lower half of the ranks.  Every unordered pair appears
    exactly once over the ranks; a rank touches at most world//2 + 1 antennas.  The first index may be the
    larger one: ``canonical_pair`` turns (j, i, lag) into (i, j, -lag)."""
    out = []
    for d in range(1, (world - 1) // 2 + 1):
        out.append((rank, (rank + d) % world))
    if world % 2 == 0 and world > 1 and rank < world // 2:
        out.append((rank, rank + world // 2))
    return out


def pair_capacity(world: int, n_ant: int) -> int:
    """Pair entries a result vector reserves: all pairs when one rank solves them, world//2 otherwise."""
    return n_ant * (n_ant - 1) // 2 if world == 1 else max(world // 2, 1)


def canonical_pair(i: int, j: int, lag: int):
    """(i, j, lag) with i < j: correlate(s_i, s_j) peaks at minus the lag of correlate(s_j, s_i)."""
    if i <= j:
        return i, j, lag
    return j, i, (lag if lag == LAG_INVALID else -lag)


def pack_stream_reference(st) -> torch.Tensor:
    """The result vector of an AntennaStream rebuilt with torch ops from the device outputs its kernels left
    (tests: gj_pack_result_dev against this packer)."""
    amp_mean = st.amp[3:4].view(torch.float32)[0]
    f = st.onset[1:3].view(torch.float32)            # noise_power, threshold, margin_hit, margin_before
    return pack_results(st.n_chunks, st.nperseg, st.power, st.stats, st.amp[0], st.amp[1], amp_mean, st.onset[0],
                        torch.tensor(0 if st.rank == 0 else LAG_INVALID), torch.tensor(0.0), f[0],
                        st.psd[:st.rows].mean(dim=0), st.rows, st.rank, pairs=st.pairs, pair_lags=st.lags,
                        pair_peaks=st.peaks, pair_margins=st.margins, capacity=st.pair_cap,
                        onset_margins=(float(f[2]), float(f[3])), onset_guard=st.onset[3], onset_threshold=float(f[1]),
                        amp_sum=float(st.amp[2:3].view(torch.float64)[0]), onset_record=st.onset, amp_record=st.amp)


def pack_results(n_chunks: int, nperseg: int, power_map: torch.Tensor, stats: torch.Tensor,
                 amp_first: torch.Tensor, amp_count: torch.Tensor, amp_mean: torch.Tensor,
                 onset: torch.Tensor, lag: torch.Tensor, peak: torch.Tensor,
                 noise_power: torch.Tensor, mean_spectrum: torch.Tensor, n_rows: int,
                 rank: int, pairs: Sequence[Tuple[int, int]] = (), pair_lags: Optional[torch.Tensor] = None,
                 pair_peaks: Optional[torch.Tensor] = None, pair_margins: Optional[torch.Tensor] = None,
                 capacity: int = 0, onset_margins: Sequence[float] = (1.0, 1.0), onset_guard: Optional[torch.Tensor] = None,
                 onset_threshold: float = 0.0, amp_sum: float = 0.0, onset_record: Optional[torch.Tensor] = None,
                 amp_record: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float64 vector [HEADER + n_chunks + nperseg + 5 capacity] (the layout of gj_pack_result_dev) built
    with device-side ops only (no host synchronisation).  int64 scalars are exact in float64 up to 2^53."""
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
    head[14] = len(pairs)
    head[15] = capacity
    head[16], head[17] = float(onset_margins[0]), float(onset_margins[1])
    head[18] = onset.to(torch.float64) if onset_guard is None else onset_guard.to(torch.float64)
    head[19] = onset_threshold
    head[20] = rank
    head[22] = 1
    head[26] = float(amp_sum)
    if onset_record is not None:          # the 32-byte device records as they are (int64[4] views)
        head[32:36] = onset_record.view(torch.float64)
    if amp_record is not None:
        head[36:40] = amp_record.view(torch.float64)
    block = torch.zeros((capacity, PAIR_FIELDS), dtype=torch.float64, device=dev)
    if len(pairs):
        block[:len(pairs), 0:2] = torch.tensor(list(pairs), dtype=torch.float64, device=dev)
        block[:len(pairs), 2] = pair_lags.to(torch.float64)[:len(pairs)]
        block[:len(pairs), 3] = pair_peaks.to(torch.float64)[:len(pairs)]
        block[:len(pairs), 4] = pair_margins.to(torch.float64)[:len(pairs)]
    return torch.cat([head, power_map.to(torch.float64), mean_spectrum.to(torch.float64), block.reshape(-1)])


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
    lag: int               # lag of this antenna relative to antenna 0 (pair (0, rank)); 0 for rank 0
    peak: float
    noise_power: float
    mean_spectrum: np.ndarray
    solved: List[Tuple[int, int, int, float, float]] = field(default_factory=list)   # (i, j, lag, peak, margin) this stream solved
    onset_margin_hit: float = 1.0      # gj_onset margins: how clearly K4's crossing cleared the threshold ...
    onset_margin_before: float = 1.0   # ... and how clearly everything in front of it stayed below
    onset_guard: int = -1              # first index inside K4's rounding band (== onset when the decision is clear)
    onset_threshold: float = 0.0

    @property
    def onset_near_tie(self) -> bool:
        """The onset was decided inside the rounding band of the reference's float32 arithmetic
        (skrypty/triangulateTDOA.py:37-49): a caller that needs the reference's exact index re-evaluates its
        expression from ``onset_guard`` on (the drop-in skrypty/triangulateTDOA.py does)."""
        if self.onset_guard != self.onset:
            return True
        return self.onset >= 0 and self.onset_margin_hit < ONSET_NEAR_TIE

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


@dataclass
class TdoaResult:
    """Every antenna pair solved on rank 0: lags[p] = lag of antenna pairs[p][1] relative to
    antenna pairs[p][0] in samples (LAG_INVALID when one of the two slots was invalid)."""
    pairs: List[Tuple[int, int]] = field(default_factory=list)
    lags: List[int] = field(default_factory=list)
    peaks: List[float] = field(default_factory=list)
    margins: List[float] = field(default_factory=list)

    def lag(self, i: int, j: int) -> int:
        return self.lags[self.pairs.index((i, j))]

    @property
    def near_ties(self) -> List[Tuple[int, int]]:
        """Pairs whose arg-max was decided inside the rounding of a complex64 FFT (margin < LAG_NEAR_TIE):
        the reference's own choice between the two lags depends on ITS rounding (triangulateTDOA.py:86-89)."""
        return [p for p, lag, m in zip(self.pairs, self.lags, self.margins) if lag != LAG_INVALID and m < LAG_NEAR_TIE]


def unpack_results(vec: torch.Tensor) -> StreamResult:
    v = vec.detach().to("cpu", torch.float64).numpy()
    n_chunks, nperseg = int(v[0]), int(v[12])
    pm = v[HEADER:HEADER + n_chunks].astype(np.float32)
    spec = v[HEADER + n_chunks:HEADER + n_chunks + nperseg].astype(np.float32)
    n_pairs = int(v[14])
    blk = v[HEADER + n_chunks + nperseg:HEADER + n_chunks + nperseg + PAIR_FIELDS * n_pairs].reshape(n_pairs, PAIR_FIELDS)
    solved = [(int(r[0]), int(r[1]), int(r[2]), float(r[3]), float(r[4])) for r in blk]
    return StreamResult(rank=int(v[13]), power_map=pm, baseline=float(v[1]), threshold=float(v[2]),
                        n_above=int(v[3]), amp_first=int(v[4]), amp_count=int(v[5]),
                        amp_mean=float(v[6]), onset=int(v[7]), lag=int(v[8]), peak=float(v[9]),
                        noise_power=float(v[10]), mean_spectrum=spec, solved=solved,
                        onset_margin_hit=float(v[16]), onset_margin_before=float(v[17]), onset_guard=int(v[18]),
                        onset_threshold=float(v[19]))


def make_slot(capture_u8: torch.Tensor, start: int, n_samples: int) -> torch.Tensor:
    """TDOA slot of one capture with torch ops only (the CPU twin of gj_tdoa_slot_dev for the
    gloo rehearsal of the exchange; the GPU pipeline uses the kernel)."""
    slot = torch.zeros(slot_bytes(n_samples), dtype=torch.uint8, device=capture_u8.device)
    ok = start >= 0 and 2 * (start + n_samples) <= capture_u8.numel()
    head = torch.tensor([0 if ok else -1, int(start)], dtype=torch.int64).view(torch.uint8)
    slot[:SLOT_HEADER] = head.to(slot.device)
    if ok:
        slot[SLOT_HEADER:SLOT_HEADER + 2 * n_samples] = capture_u8[2 * start:2 * (start + n_samples)]
    return slot


def slot_fields(slot: torch.Tensor, n_samples: int):
    """(valid, start, uint8 slice) of one slot (host side; tests)."""
    raw = slot.detach().cpu().contiguous()
    flag, start = raw[:SLOT_HEADER].view(torch.int64).tolist()
    return flag == 0, start, raw[SLOT_HEADER:SLOT_HEADER + 2 * n_samples].numpy()


def gather_rows(row: torch.Tensor, rank: int, world_size: int, dst: int = 0,
                out: Optional[torch.Tensor] = None, always: bool = False, group=None) -> Optional[torch.Tensor]:
    """Every rank's ``row`` on ``dst`` as one [world, len] tensor (rows in rank order); None on the
    other ranks.  One collective (torch.distributed.gather: RCCL on HIP tensors, gloo on CPU).
    ``always``: issue the collective even in a group of one (tests / bench.py --force-exchange: the RCCL
    call path on a single GPU).  ``group``: the process group that carries the data path (None = the default group):
    bench.py keeps a gloo group for control and votes and gives the exchange an RCCL group of its own."""
    if world_size == 1 and not always:
        return row.unsqueeze(0)
    import torch.distributed as dist
    if rank == dst:
        if out is None:
            out = torch.empty((world_size, row.numel()), dtype=row.dtype, device=row.device)
        dist.gather(row, gather_list=[out[r] for r in range(world_size)], dst=dst, group=group)
        return out
    dist.gather(row, gather_list=None, dst=dst, group=group)
    return None


def allgather_rows(row: torch.Tensor, world_size: int, out: Optional[torch.Tensor] = None,
                   always: bool = False, group=None) -> torch.Tensor:
    """Every rank's ``row`` on EVERY rank as one [world, len] tensor: ONE collective straight into the
    contiguous receive buffer (all_gather_into_tensor = ncclAllGather on HIP tensors; the list form of
    all_gather flattens and copies out once more).  ``always``: as in ``gather_rows``."""
    if world_size == 1 and not always:
        return row.unsqueeze(0)
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world_size, row.numel()), dtype=row.dtype, device=row.device)
    assert out.is_contiguous() and out.shape == (world_size, row.numel()) and out.dtype == row.dtype
    dist.all_gather_into_tensor(out.view(-1), row.contiguous().view(-1), group=group)   # flat form: accepted by nccl and gloo alike
    return out


def gather_results(vec: torch.Tensor, rank: int, world_size: int, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather every rank's result vector on ``dst``; returns the list there, None elsewhere."""
    rows = gather_rows(vec, rank, world_size, dst)
    return None if rows is None else [rows[r] for r in range(rows.shape[0])]


class StepResults:
    """What ``AntennaStream.exchange`` returns on rank 0: the gathered result vectors (with the pair
    tables inside), still in HBM, plus the event that marks them complete.  The collectives and K5 run
    on the pipeline's second stream; a consumer on any other stream must call ``wait()`` (or go through
    ``unpack()`` / indexing, which do) before reading the tensors.  The buffers are reused two steps
    later."""

    def __init__(self, vectors, event, n_ant):
        self.vectors, self.event, self.n_ant = vectors, event, n_ant
        self.near_ties = None      # filled by unpack(): {"onset": [ranks], "lag": [pairs]} decided inside a rounding band

    def wait(self, stream=None):
        """Make ``stream`` (default: torch's current stream) wait for the exchange."""
        if self.event is not None:
            stream = stream or torch.cuda.current_stream(self.vectors.device)
            stream.wait_event(self.event)
            if self.vectors.is_cuda:
                self.vectors.record_stream(stream)
        return self

    def __len__(self):
        return self.vectors.shape[0]

    def __getitem__(self, r):
        self.wait()
        return self.vectors[r]

    def unpack(self) -> Tuple[List[StreamResult], TdoaResult]:
        """Host-side view: one StreamResult per rank (lag / peak = pair (0, rank)) and every solved pair,
        in canonical order (i < j, sorted)."""
        self.wait()
        res = [unpack_results(self.vectors[r]) for r in range(self.vectors.shape[0])]
        table = {}
        for r in res:
            for i, j, lag, peak, margin in r.solved:
                ci, cj, clag = canonical_pair(i, j, lag)
                table[(ci, cj)] = (clag, peak, margin)
        keys = sorted(table)
        td = TdoaResult(keys, [table[k][0] for k in keys], [table[k][1] for k in keys], [table[k][2] for k in keys])
        for r in res:
            if r.rank == 0:
                r.lag, r.peak = 0, 0.0
            elif (0, r.rank) in table:
                r.lag, r.peak = table[(0, r.rank)][0], table[(0, r.rank)][1]
        # decisions taken inside a rounding band: rank 0 knows, and says so once per step
        self.near_ties = {"onset": [r.rank for r in res if r.onset_near_tie], "lag": td.near_ties}
        if self.near_ties["onset"] or self.near_ties["lag"]:
            _log.warning("near-tie decisions: onset of stream(s) %s, lag of pair(s) %s -- decided by the GPU's exact "
                         "arithmetic; the reference's float32 rounding could choose a neighbouring index",
                         self.near_ties["onset"], self.near_ties["lag"])
        return res, td

    def tdoa(self) -> TdoaResult:
        return self.unpack()[1]


class AntennaStream:
    """Device-resident pipeline of one capture on one GPU (uses torch only for device
    memory, streams and -- with ``transport="torch"`` -- the collectives; every kernel is a
    gpsjam C-ABI call).

    ``exchange_always``: with ``world_size == 1`` and an initialised process group of one, still issue the slot
    all-gather and the result gather (what N > 1 ranks do), so that the collective path can be run on one GPU.
    ``pairs`` (single-rank use): solve only these antenna pairs instead of all of them; ``side_priority``: HIP stream
    priority of the second stream (0 default, -1 high).
    ``aux_slots`` (single-rank use): pre-filled TDOA slots of further antennas, [n_aux, slot_bytes]
    uint8; the rank then solves all pairs over 1 + n_aux antennas itself (BASELINE configs[3]:
    three antennas, three pairs on one GPU).
    ``transport``: "torch" = torch.distributed (nccl = RCCL on HIP tensors, gloo for rehearsal);
    "rccl" = the library's own gj_comm_* entry points (gpsjam.comm), no torch.distributed; or a
    ready ``gpsjam.comm.Communicator``.  The collectives are enqueued on the stream of the context the
    communicator was made on, and that must be the context the slot and K5 kernels run on -- with
    ``overlap`` the side context: pass it in as ``side_device`` (and build the Communicator on it);
    a communicator bound to any other context is refused."""

    def __init__(self, dev, capture: torch.Tensor, *, chunk_bytes: int = 65536,
                 chunk_samples: int = 2048000, nperseg: int = 4096, fs: float = 2.048e6,
                 slice_samples: int = 1 << 19, noise_samples: int = 200000, window: int = 1000,
                 factor: float = 50.0, rssi_threshold: float = 0.0, rank: int = 0, world_size: int = 1,
                 overlap: Optional[bool] = None, aux_slots: Optional[torch.Tensor] = None,
                 transport="torch", side_device=None, exchange_always: bool = False, pairs=None,
                 side_priority: int = 0, pack_on_side: bool = False, group=None):
        assert capture.dtype == torch.uint8 and capture.is_contiguous()
        self.dev, self.cap = dev, capture
        self.group = group                 # torch.distributed group of the exchange (None: the default group)
        # K2 is bound by VALU issue and leaves ~90 % of the HBM bandwidth idle, the fused scan is HBM
        # bound: with ``overlap`` the scan, threshold and TDOA kernels run on a second HIP
        # stream (own gpsjam context = own workspace) concurrently with K2 and join in pack().
        if overlap is None:
            overlap = capture.is_cuda and hasattr(dev, "_ctx")
        self.overlap = bool(overlap)
        self.streams_overlap = None        # one stream: nothing to overlap
        self.dev_side = dev
        self._main = None
        if capture.is_cuda and hasattr(dev, "_ctx"):
            # the pipeline's torch ops, its events and the gpsjam kernels must share one stream
            self._main = torch.cuda.current_stream(capture.device)
            dev.set_stream(self._main.cuda_stream)
        self._own_side = False
        if self.overlap:
            self.dev_side = side_device if side_device is not None else type(dev)(dev.index)
            self._own_side = side_device is None
            # side_priority < 0: a high-priority HIP stream -- its (short) kernels are dispatched ahead of K2's waiting
            # workgroups, which shortens the scan -> slot -> exchange -> K5 chain without changing the total work
            # ... on a hardware queue of its own: two streams the runtime has mapped to one queue run one after the other
            from .streams import stream_beside_checked
            #: False: no stream could be found that runs beside the main one (results unaffected, chains serialised)
            self._side, self.streams_overlap = stream_beside_checked([(dev, self._main)], device=capture.device, priority=int(side_priority))
            self.dev_side.set_stream(self._side.cuda_stream)
            self._ev_free = torch.cuda.Event()      # main: previous results consumed, buffers may be rewritten
            self._ev_side = torch.cuda.Event()      # side: scan / TDOA results ready
            self._ev_packed = torch.cuda.Event()    # main: this step's result vector is packed
            self._ev_free.record(self._main)
        else:
            self._side = self._main
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
        # ``pack_on_side``: the result vector is packed on the second stream (behind K5, where its inputs come from), so the
        # main stream carries K2 + finalize and nothing else; needs two PSD buffers, written alternately (gpsjam/split.py
        # does the same, where it pays: NOTES_r05 section 5).  Off by default: on the 1-GiB step it is measured neutral.
        self._pack_on_side = bool(pack_on_side) and self.overlap
        self.psd2 = [torch.empty((max(self.rows, 1), nperseg), dtype=torch.float32, device=d) for _ in range(2 if self._pack_on_side else 1)]
        self.psd = self.psd2[0]
        self._pidx = 0
        if self._pack_on_side:
            self._ev_psd = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev_psd_read = [None, None]
        self.amp = torch.zeros(4, dtype=torch.int64, device=d)        # gj_amp_stats (32 bytes)
        self.onset = torch.zeros(4, dtype=torch.int64, device=d)      # gj_onset (32 bytes)
        # collectives
        self.comm = None
        if transport == "rccl" and world_size > 1:
            from .comm import Communicator
            self.comm = Communicator(self.dev_side, rank, world_size)
        elif not isinstance(transport, str):
            if getattr(transport, "dev", None) is not self.dev_side:
                raise ValueError("the Communicator must be made on the context the exchange runs on"
                                 + (" (pass that Device as side_device=)" if self.overlap else ""))
            self.comm = transport
        # a communicator is used even when alone; ``exchange_always`` does the same for torch.distributed (a process
        # group of one: every collective of the N > 1 path is issued, on the same streams, on a single GPU)
        self._always = bool(exchange_always) and self.comm is None
        self._exchange = world_size > 1 or self.comm is not None or self._always
        # TDOA: the slots of every antenna (all-gathered: every rank holds them all), the pairs THIS rank solves
        self.slot_bytes = dev.tdoa_slot_bytes(slice_samples)
        n_aux = 0 if aux_slots is None else int(aux_slots.shape[0])
        assert not (n_aux and world_size > 1), "aux_slots is the single-rank form"
        self.n_ant = world_size if world_size > 1 else 1 + n_aux
        self.is_root = rank == 0
        self.pairs = all_pairs(self.n_ant) if world_size == 1 else pairs_of_rank(rank, world_size)
        if pairs is not None:                      # single-rank use: solve these pairs only (load rehearsals)
            assert world_size == 1 and all(0 <= i < self.n_ant and 0 <= j < self.n_ant for i, j in pairs)
            self.pairs = [tuple(p) for p in pairs]
        self.pair_cap = max(pair_capacity(world_size, self.n_ant), len(self.pairs))
        self.slots = torch.zeros((self.n_ant, self.slot_bytes), dtype=torch.uint8, device=d)
        if n_aux:
            assert aux_slots.shape[1] == self.slot_bytes and aux_slots.dtype == torch.uint8
            self.slots[1:].copy_(aux_slots)
        # no exchange: the slot is written in place; otherwise a send buffer of its own
        self.my_slot = self.slots[0] if not self._exchange else torch.zeros(self.slot_bytes, dtype=torch.uint8, device=d)
        npairs = max(len(self.pairs), 1)
        self.d_pairs = torch.tensor([x for p in self.pairs for x in p] or [0, 0], dtype=torch.int32, device=d)
        self.lags = torch.full((npairs,), LAG_INVALID, dtype=torch.int32, device=d)
        self.peaks = torch.zeros(npairs, dtype=torch.float32, device=d)
        self.margins = torch.zeros(npairs, dtype=torch.float32, device=d)
        # two sets, used alternately: step k's consumer may still be reading one while step k + 1 fills the other
        rl = result_len(self.n_chunks, nperseg, self.pair_cap)
        self._results = [torch.zeros(rl, dtype=torch.float64, device=d) for _ in range(2)]
        self._gathered = ([torch.zeros((world_size, rl), dtype=torch.float64, device=d) for _ in range(2)]
                          if (self.is_root and self._exchange) else [None, None])
        self._done = [torch.cuda.Event() for _ in range(2)] if capture.is_cuda else [None, None]
        self._idx = 0
        self.result = self._results[0]
        # workspaces: K5 transforms at most the antennas this rank's pairs name
        ants = len({a for p in self.pairs for a in p}) or 1
        ws_side = max(dev.xcorr_workspace(ants, slice_samples, npairs), self.nbytes // 48 + (1 << 20))
        ws_main = dev.welch_workspace(self.nbytes, chunk_samples, nperseg)
        if self.overlap:
            dev.reserve(ws_main)
            self.dev_side.reserve(ws_side)
        else:
            dev.reserve(max(ws_main, ws_side))

    def _on_side(self):
        """Context manager: torch's current stream = the side stream (no-op without overlap)."""
        return torch.cuda.stream(self._side) if self.overlap else contextlib.nullcontext()

    # ---------------------------------------------------------------- per-capture kernels
    def stream_scan(self):
        """K1 + K3 + K4 in one pass over the capture, then the noise-floor threshold."""
        if self.overlap:
            self._side.wait_event(self._ev_free)
        d = self.dev_side
        # two launches: the fused pass, then the tail -- threshold, amplitude totals, onset record and this capture's
        # TDOA slot (gj_capture_scan_dev; tdoa() finds the slot cut)
        d.capture_scan_dev(self.cap, self.nbytes, self.chunk_bytes, self.power, self.rssi_threshold,
                           self.amp, self.noise_samples, self.window, self.factor, self.onset,
                           d_stats=self.stats if self.n_chunks else None, d_mask=self.mask if self.n_chunks else None,
                           slice_samples=self.slice_samples, d_slot=self.my_slot)
        self._slot_cut = True
        if self.overlap:
            self._ev_side.record(self._side)

    def welch(self):
        if self._pack_on_side:
            self._pidx ^= 1
            self.psd = self.psd2[self._pidx]
            if self._ev_psd_read[self._pidx] is not None:           # the packing of two steps ago has read this buffer
                self._main.wait_event(self._ev_psd_read[self._pidx])
        self.dev.welch_dev(self.cap, self.nbytes, self.chunk_samples, self.nperseg, self.fs, self.psd)
        if self._pack_on_side:
            self._ev_psd[self._pidx].record(self._main)

    def scan(self):
        """Everything that only needs this rank's capture (no host synchronisation)."""
        self.stream_scan()
        self.welch()

    # ---------------------------------------------------------------- the TDOA exchange
    def tdoa(self):
        """Own slice -> TDOA slot; ONE all-gather puts every rank's slot on every rank; this rank solves
        its share of the antenna pairs with one multi-pair K5 launch.  All on the second stream, beside K2."""
        with self._on_side():
            dev = self.dev_side
            if not getattr(self, "_slot_cut", False):      # tdoa() without a stream_scan() in front of it
                dev.tdoa_slot_dev(self.cap, self.nbytes, self.onset, self.slice_samples, self.my_slot)
            self._slot_cut = False
            if self._exchange:
                if self.comm is not None:
                    self.comm.allgather(self.my_slot, self.slot_bytes, self.slots)
                else:
                    allgather_rows(self.my_slot, self.world, out=self.slots[:self.world], always=self._always, group=self.group)
            if self.pairs:
                dev.xcorr_slots_dev(self.slots, self.slot_bytes, self.n_ant, self.slice_samples, self.pairs,
                                    self.lags, self.peaks, self.margins)
            if self.overlap:
                self._ev_side.record(self._side)

    def pack(self) -> torch.Tensor:
        """Result vector of this stream, built by one kernel (layout = pack_results)."""
        side = self._pack_on_side
        stream, dev = (self._side, self.dev_side) if side else (self._main, self.dev)
        if side:
            stream.wait_event(self._ev_psd[self._pidx])
        elif self.overlap:
            self._main.wait_event(self._ev_side)
        self._idx ^= 1
        self.result = self._results[self._idx]
        dev.pack_result_dev(self.n_chunks, self.power, self.stats, self.amp, self.onset, self.psd, self.rows,
                            self.nperseg, self.rank, len(self.pairs), self.pair_cap, self.d_pairs, self.lags,
                            self.peaks, self.margins, self.result)
        if self.overlap:
            self._ev_free.record(stream)       # the next step's scan / K5 may overwrite their outputs now
            self._ev_packed.record(stream)
            if side:
                if self._ev_psd_read[self._pidx] is None:
                    self._ev_psd_read[self._pidx] = torch.cuda.Event()
                self._ev_psd_read[self._pidx].record(stream)
        return self.result

    def exchange(self, dst: int = 0) -> Optional[StepResults]:
        """Pack this step's result vector and gather every rank's on rank 0.  With two streams the
        collective is issued on the second one (after the packing kernel), so the main stream goes
        straight on to the next step's K2 instead of waiting for the gather; the next step's scan
        follows the gather in stream order.  The returned StepResults carries the event a consumer
        on any other stream has to wait for (``wait()`` / ``unpack()``); buffers are reused two
        steps later."""
        assert dst == 0, "rank 0 collects"
        vec = self.pack()
        k = self._idx
        if not self._exchange:
            rows = vec.unsqueeze(0)
            if self._done[k] is not None:
                self._done[k].record(self._side if self._pack_on_side else self._main)
        else:
            if self.overlap:
                self._side.wait_event(self._ev_packed)
            with self._on_side():
                if self.comm is not None:
                    rows = self._gathered[k]
                    self.comm.gather(vec, vec.numel() * vec.element_size(), rows if self.is_root else None, 0)
                else:
                    rows = gather_rows(vec, self.rank, self.world, 0, out=self._gathered[k], always=self._always, group=self.group)
                if self._done[k] is not None:
                    self._done[k].record(self._side)
        if not self.is_root:
            return None
        return StepResults(rows, self._done[k], self.n_ant)

    def step(self) -> Optional[StepResults]:
        """One pass of the hot path over this rank's capture + the exchange."""
        self.scan()
        self.tdoa()
        return self.exchange(0)

    def close(self):
        if self.comm is not None and hasattr(self.comm, "close"):
            self.comm.close()
        if self.overlap and self._own_side and self.dev_side is not self.dev:
            self.dev_side.close()
