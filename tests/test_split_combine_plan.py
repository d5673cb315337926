"""Host logic of rank 0's three-launch combine (gpsjam/split.py::_build_combine, include/gpsjam.h gj_combine_plan_create):
the static COPY LIST -- which run of which part vector goes where in which capture-order array -- checked on the CPU.
A recording stand-in takes the place of the device; the part vectors are filled with values that say where they came
from (antenna, kind, index in the capture); the copy list is then executed in numpy exactly as combine_assemble_kernel
executes it, and every assembled array must come out in capture order.  No GPU, no kernels: pure indexing."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "gps-jamming_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

from gpsjam import _ffi, split  # noqa: E402
from gpsjam.sharded import HEADER, PAIR_FIELDS  # noqa: E402


class _RecordingDevice:
    """What SplitStreams asks of a device at construction -- sizes from the library's host-side helpers (no GPU needed),
    workspaces of nothing -- and a combine_plan that keeps what it is given."""
    index = 0

    def __init__(self):
        self._lib = _ffi.load()
        self.plans = []

    def chunk_count(self, n, c):
        return self._lib.gj_chunk_count(n, c)

    def welch_rows(self, n, cs, nper):
        return self._lib.gj_welch_rows(n, cs, nper)

    def amp_tile_count(self, n):
        return self._lib.gj_amp_tile_count(int(n))

    def part_result_len(self, *a):
        return self._lib.gj_part_result_len(*a)

    def part_welch_workspace(self, *a):
        return 0

    def xcorr_workspace(self, *a):
        return 0

    def reserve(self, n):
        pass

    def set_stream(self, *a, **k):
        pass

    def combine_plan(self, copies, captures, rows_bytes, arena, nperseg, d_pairs, d_lags, d_peaks, d_margins, pct=5.0, rise_db=6.0):
        self.plans.append(dict(copies=list(copies), captures=list(captures), rows_bytes=rows_bytes, arena=arena, nperseg=nperseg,
                               lags=d_lags, peaks=d_peaks, margins=d_margins))
        return len(self.plans)

    def combine_plan_destroy(self, plan):
        pass

    def close(self):
        pass


def _execute(plan, rows: np.ndarray):
    """combine_assemble_kernel in numpy: every copy of the list, on a host image of the arena."""
    arena = np.zeros(plan["arena"].numel(), np.uint8)
    base = plan["arena"].data_ptr()
    raw = rows.view(np.uint8).reshape(-1)
    assert raw.size == plan["rows_bytes"]
    for c in plan["copies"]:
        src_t = np.float32 if c.kind == _ffi.GJ_COPY_F32 else np.float64
        dst_t = {_ffi.GJ_COPY_F64_F32: np.float32, _ffi.GJ_COPY_F64: np.float64, _ffi.GJ_COPY_F32: np.float32,
                 _ffi.GJ_COPY_F64_I32: np.int32}[c.kind]
        esz = np.dtype(src_t).itemsize
        assert c.src_byte % esz == 0 and c.src_stride % esz == 0 and c.src_byte + (c.count - 1) * c.src_stride + esz <= raw.size
        idx = c.src_byte // esz + np.arange(c.count) * (c.src_stride // esz)
        vals = raw.view(src_t)[idx].astype(dst_t)
        o = c.dst - base
        assert 0 <= o and o + vals.nbytes <= arena.size and o % np.dtype(dst_t).itemsize == 0
        assert not arena[o:o + vals.nbytes].any(), "two copies write the same place"
        arena[o:o + vals.nbytes] = vals.view(np.uint8)
    return arena, base


@pytest.mark.parametrize("sizes,world", [([40_960_000] * 3, 8), ([3_400_000, 3_387_655, 3_400_000], 4), ([1 << 26], 3),
                                          ([20_000_000, 9_000_000], 5)])
def test_copy_list_rebuilds_every_capture_in_order(sizes, world):
    chunk_samples = 2048000 if max(sizes) > 16_000_000 else 131072
    nper, slice_samples = 1024, 50000
    dev = _RecordingDevice()
    buffers = []

    def make_buffer(part, b0, b1):
        buffers.append(torch.zeros(b1 - b0, dtype=torch.uint8))
        return buffers[-1]

    def make_noise(antenna, n):
        return torch.zeros(n, dtype=torch.uint8)

    st = split.SplitStreams(dev, sizes, make_buffer, make_noise, rank=0, world_size=world, chunk_samples=chunk_samples,
                            nperseg=nper, slice_samples=slice_samples, device=torch.device("cpu"), overlap=False)
    assert len(dev.plans) == 2 and st.combine_launches == 3                # one plan per arena, three launches a step
    L = st.part_len
    # part vectors that say where their contents belong: value = 1000 * kind + antenna * 1e7 + index in the CAPTURE
    rows = np.zeros((world, st.pmax * L), np.float64)
    want = {}
    for p in st.parts:
        v = rows[p.rank, p.local * L:(p.local + 1) * L]
        nc, nt = dev.chunk_count(p.own_bytes, 65536), dev.amp_tile_count(p.own_bytes)
        nr = dev.welch_rows(p.own_bytes, chunk_samples, nper)
        c0, t0, r0 = p.first_byte // 65536, p.first_byte // 65536, p.first_byte // (2 * chunk_samples)
        tag = p.antenna * 1e7
        v[HEADER:HEADER + nc] = 1e3 + tag + np.arange(c0, c0 + nc)
        v[st.o_tiles:st.o_tiles + 2 * nt] = 2e3 + tag + np.arange(2 * t0, 2 * (t0 + nt))
        v[32:36] = 3e3 + tag + 4 * p.part + np.arange(4)
        v[36:40] = 4e3 + tag + 4 * p.part + np.arange(4)
        v[st.o_rows:].view(np.float32)[:nr * nper] = (5e3 + p.antenna * 1e5 + np.arange(r0 * nper, (r0 + nr) * nper) % 65536).astype(np.float32)
    q = 0
    for r in sorted(st.deal):                                                  # pairs ride on a rank's FIRST part
        for k, (i, j) in enumerate(st.deal[r]):
            blk = rows[r, st.o_pairs + PAIR_FIELDS * k:st.o_pairs + PAIR_FIELDS * (k + 1)]
            blk[:] = [i, j, 100 + q, 0.5 + q, 0.25 + q]
            want[q] = (100 + q, 0.5 + q, 0.25 + q)
            q += 1
    for plan in dev.plans:
        arena, base = _execute(plan, rows)
        for a, cap in enumerate(plan["captures"]):
            n_chunks, n_rows, n_tiles = cap.n_chunks, cap.rows, cap.n_tiles
            assert (n_chunks, n_rows, n_tiles, cap.total_bytes) == (dev.chunk_count(sizes[a], 65536), dev.welch_rows(sizes[a], chunk_samples, nper),
                                                                      dev.amp_tile_count(sizes[a]), sizes[a])
            tag = a * 1e7
            f32 = lambda ptr, n: arena[ptr - base:ptr - base + 4 * n].view(np.float32)      # noqa: E731
            f64 = lambda ptr, n: arena[ptr - base:ptr - base + 8 * n].view(np.float64)      # noqa: E731
            np.testing.assert_array_equal(f32(cap.d_power, n_chunks), (1e3 + tag + np.arange(n_chunks)).astype(np.float32))
            np.testing.assert_array_equal(f64(cap.d_tiles, 2 * n_tiles), 2e3 + tag + np.arange(2 * n_tiles))
            parts_a = [p for p in st.parts if p.antenna == a]
            assert cap.n_parts == len(parts_a) and cap.antenna == a
            np.testing.assert_array_equal(f64(cap.d_onset_parts, 4 * cap.n_parts), 3e3 + tag + np.arange(4 * cap.n_parts))
            np.testing.assert_array_equal(f64(cap.d_amp_parts, 4 * cap.n_parts), 4e3 + tag + np.arange(4 * cap.n_parts))
            np.testing.assert_array_equal(f32(cap.d_psd, n_rows * nper),
                                          (5e3 + a * 1e5 + np.arange(n_rows * nper) % 65536).astype(np.float32))
            assert cap.n_pairs == (q if a == 0 else 0) and cap.pair_cap == st.total_pairs
        lags = arena[plan["lags"] - base:plan["lags"] - base + 4 * max(q, 1)].view(np.int32)
        peaks = arena[plan["peaks"] - base:plan["peaks"] - base + 4 * max(q, 1)].view(np.float32)
        margs = arena[plan["margins"] - base:plan["margins"] - base + 4 * max(q, 1)].view(np.float32)
        for k in range(q):
            assert (lags[k], peaks[k], margs[k]) == (want[k][0], np.float32(want[k][1]), np.float32(want[k][2]))
        # the pairs in the order of the static pair table rank 0 packs beside them
        assert st._d_all_pairs.tolist()[:2 * q] == [x for r in sorted(st.deal) for pr in st.deal[r] for x in pr]
    st.close()


# ----------------------------------------------------------------------------- the plan's bounds must not wrap
def _plan_lists(arena, n_chunks=8, nperseg=1024):
    """One capture whose arrays sit in `arena` (a numpy buffer standing in for device memory: the check is address
    arithmetic only) + one honest copy into its power map."""
    base = arena.ctypes.data
    off = iter(range(0, 1 << 20, 1 << 12))
    cap = _ffi.CombineCapture(n_chunks=n_chunks, rows=1, n_tiles=n_chunks, total_bytes=n_chunks * 65536, n_parts=2, antenna=0,
                              n_pairs=0, pair_cap=3, d_power=base + next(off), d_stats=base + next(off), d_tiles=base + next(off),
                              d_amp_parts=base + next(off), d_onset_parts=base + next(off), d_amp=base + next(off),
                              d_onset=base + next(off), d_psd=base + 0x10000, d_out=base + 0x20000)
    copy = _ffi.CombineCopy(src_byte=40 * 8, dst=cap.d_power, count=n_chunks, src_stride=8, kind=_ffi.GJ_COPY_F64_F32)
    return cap, copy


def _check(lib, copies, caps, rows_bytes, arena, nperseg=1024):
    cs = (_ffi.CombineCopy * len(copies))(*copies)
    ca = (_ffi.CombineCapture * len(caps))(*caps)
    return lib.gj_combine_plan_check(cs, len(copies), ca, len(caps), rows_bytes, arena.ctypes.data, arena.nbytes, nperseg, 0)


def test_combine_plan_bounds_are_taken_by_division():
    """gj_combine_plan_create promises that a bad copy or descriptor never reaches the GPU.  Its fields are caller-supplied
    64-bit numbers: a count of 2^61 makes count * 8 wrap to 0, a count of 2^61 + 1 with stride 8 makes the last source
    byte wrap to a small number -- sums and products pass where the honest bound does not.  gj_combine_plan_check is the
    create call's validation alone (host arithmetic, no GPU)."""
    lib = _ffi.load()
    arena = np.zeros(1 << 20, np.uint8)
    rows_bytes = 4096
    cap, copy = _plan_lists(arena)
    assert _check(lib, [copy], [cap], rows_bytes, arena) == 0                       # the honest plan passes
    bad = []
    for count, stride in ((1 << 61, 8), ((1 << 61) + 1, 8), (1 << 63, 8), ((1 << 64) - 1, 8), (1 << 32, 1 << 31), (513, 8)):
        c = _ffi.CombineCopy(src_byte=copy.src_byte, dst=copy.dst, count=count, src_stride=stride, kind=copy.kind)
        bad.append(("count %d stride %d" % (count, stride), [c], [cap]))
    c = _ffi.CombineCopy(src_byte=(1 << 64) - 8, dst=copy.dst, count=2, src_stride=8, kind=copy.kind)   # src_byte + stride wraps
    bad.append(("src_byte at the top of the address space", [c], [cap]))
    c = _ffi.CombineCopy(src_byte=0, dst=arena.ctypes.data + arena.nbytes - 4, count=2, src_stride=8, kind=copy.kind)
    bad.append(("destination runs off the arena", [c], [cap]))
    for field, value in (("n_chunks", 1 << 62), ("n_chunks", (1 << 64) - 40), ("rows", 1 << 61), ("n_parts", 1 << 30),
                         ("pair_cap", (1 << 31) - 1)):
        cap2, _ = _plan_lists(arena)
        setattr(cap2, field, value)
        if field == "n_chunks":
            cap2.n_tiles, cap2.total_bytes = 8, 8 * 65536
        bad.append((f"capture {field} = {value}", [copy], [cap2]))
    for what, copies, caps in bad:
        assert _check(lib, copies, caps, rows_bytes, arena) == -1, what               # GJ_ERR_INVALID, never a wrapped "ok"
    # limits of the lists themselves
    assert lib.gj_combine_plan_check(None, 1, None, 1, 0, None, 0, 1024, 0) == -1
    assert _check(lib, [copy], [cap], rows_bytes, arena, nperseg=1000) == -5
