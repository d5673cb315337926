"""Randomised check of the capture-part entry points in ONE process (no torch, no ranks): a capture is cut at random
unit boundaries, every part is scanned / transformed / sliced with gj_part_* from a buffer that holds only its own
bytes (+ halo, tail, noise span), the combine kernels put the capture together, and every result must equal -- bit
for bit -- what the unsplit capture gives: burst edges next to cut points, onsets inside a halo, first amplitude hits
in any part, captures without an onset, ragged odd tails."""
import ctypes as C

import numpy as np
import pytest

import gpsjam
from gpsjam import _ffi
from gpsjam.synth import StreamSpec, generate

pytestmark = pytest.mark.gpu

TILE, UNIT = 65536, 65536            # chunk_bytes 65536, chunk_samples 32768: a unit is one tile
NOISE, WINDOW, SLICE, NPERSEG = 20000, 1000, 1 << 13, 256


def _whole(dev, raw, thr):
    n = raw.size
    buf = dev.alloc(n + 16).upload(raw)
    nch, rows = dev.chunk_count(n, 65536), dev.welch_rows(n, 32768, NPERSEG)
    d_pow, d_amp, d_on, d_psd = dev.alloc(4 * nch), dev.alloc(32), dev.alloc(32), dev.alloc(4 * max(rows, 1) * NPERSEG)
    d_slot = dev.alloc(dev.tdoa_slot_bytes(SLICE))
    dev.stream_scan_dev(buf, n, 65536, d_pow, thr, d_amp, NOISE, WINDOW, 50.0, d_on)
    dev.welch_dev(buf, n, 32768, NPERSEG, 2.048e6, d_psd)
    dev.tdoa_slot_dev(buf, n, d_on, SLICE, d_slot)
    dev.synchronize()
    out = (d_pow.download(np.float32, nch), d_amp.download(np.uint8, 32).tobytes(), d_on.download(np.uint8, 32).tobytes(),
           d_psd.download(np.float32, rows * NPERSEG), d_slot.download(np.uint8).tobytes())
    for b in (buf, d_pow, d_amp, d_on, d_psd, d_slot):
        b.free()
    return out


def _parts(dev, raw, cuts, thr):
    n = raw.size
    nch, rows, ntiles = dev.chunk_count(n, 65536), dev.welch_rows(n, 32768, NPERSEG), dev.amp_tile_count(n)
    sb = dev.tdoa_slot_bytes(SLICE)
    G = len(cuts) - 1
    d_pow, d_psd, d_tiles = dev.alloc(4 * nch), dev.alloc(4 * max(rows, 1) * NPERSEG), dev.alloc(16 * ntiles)
    d_amp_parts, d_on_parts, d_slots = dev.alloc(32 * G), dev.alloc(32 * G), dev.alloc(sb * G)
    d_noise = dev.alloc(2 * NOISE).upload(raw[:2 * NOISE])
    bufs = []
    for g in range(G):
        halo = TILE if g else 0
        b0 = cuts[g] - halo
        b1 = min(n, cuts[g + 1] + 2 * SLICE + TILE)
        buf = dev.alloc(b1 - b0 + 16).upload(raw[b0:b1])                 # this part's bytes only
        bufs.append(buf)
        view = _ffi.PartView(buf.ptr, b1 - b0, b0, cuts[g], cuts[g + 1] - cuts[g], n, d_noise.ptr)
        dev.part_scan_dev(view, 65536, d_pow.ptr + 4 * (cuts[g] // 65536), thr, d_tiles.ptr + 16 * (cuts[g] // TILE),
                          d_amp_parts.ptr + 32 * g, NOISE, WINDOW, 50.0, d_on_parts.ptr + 32 * g)
        dev.part_welch_dev(view, 32768, NPERSEG, 2.048e6, d_psd.ptr + 4 * NPERSEG * (cuts[g] // 65536))
        dev.part_slot_dev(view, d_on_parts.ptr + 32 * g, SLICE, d_slots.ptr + sb * g)
    d_amp, d_on, d_slot = dev.alloc(32), dev.alloc(32), dev.alloc(sb)
    d_groups = dev.alloc(4 * (2 + G)).upload(np.array([0, G] + list(range(G)), np.int32))
    dev.amp_combine_dev(d_tiles, ntiles, d_amp_parts, G, n, d_amp)
    dev.onset_combine_dev(d_on_parts, G, d_on)
    dev.slots_pick_dev(d_slots, sb, d_groups.ptr, d_groups.ptr + 8, 1, d_slot)
    dev.synchronize()
    out = (d_pow.download(np.float32, nch), d_amp.download(np.uint8, 32).tobytes(), d_on.download(np.uint8, 32).tobytes(),
           d_psd.download(np.float32, rows * NPERSEG), d_slot.download(np.uint8).tobytes())
    for b in bufs + [d_pow, d_psd, d_tiles, d_amp_parts, d_on_parts, d_slots, d_noise, d_amp, d_on, d_slot, d_groups]:
        b.free()
    return out


@pytest.mark.parametrize("seed", range(48))
def test_random_cuts_are_bit_identical(dev, seed):
    rng = np.random.default_rng(1000 + seed)
    units = int(rng.integers(6, 40))
    n = units * UNIT - int(rng.integers(0, UNIT // 2)) * int(rng.integers(0, 2))          # ragged (maybe odd) tail or none
    if n % 2 == 0 and rng.integers(0, 2):
        n -= 1
    nsamp = n // 2
    n_cuts = int(rng.integers(1, min(5, units - 1)))
    inner = sorted(rng.choice(np.arange(1, (n + UNIT - 1) // UNIT), size=n_cuts, replace=False).tolist())
    cuts = [0] + [c * UNIT for c in inner] + [n]
    kind = seed % 4
    if kind == 0:      # burst edge a few samples around a cut point (onset inside a halo / right behind a cut)
        edge = cuts[1 + int(rng.integers(0, n_cuts))] // 2 + int(rng.integers(-1200, 1200))
    elif kind == 1:    # burst early
        edge = NOISE + WINDOW + int(rng.integers(10, 5000))
    elif kind == 2:    # no burst at all: no onset, no slot
        edge = 1 << 40
    else:
        edge = int(rng.integers(NOISE + 2000, max(NOISE + 3000, nsamp - 100)))
    edge = max(edge, NOISE + 10)
    spec = StreamSpec(seed=int(rng.integers(1, 1 << 30)), jam_start=edge, jam_end=1 << 41, jam_sigma=float(rng.uniform(45, 70)))
    raw = generate(spec, nsamp + 1)[:n].copy()
    thr = float(rng.choice([0.0, 0.05, 0.3, 0.6, 2.0]))
    want = _whole(dev, raw, thr)
    got = _parts(dev, raw, cuts, thr)
    names = ("power map", "amplitude record", "onset record", "PSD rows", "TDOA slot")
    for name, g, w in zip(names, got, want):
        if name == "onset record":               # margin_before is a bound that depends on what was screened: not compared
            go, wo = _ffi.Onset.from_buffer_copy(g), _ffi.Onset.from_buffer_copy(w)
            assert (go.start_index, go.guard_index, go.noise_power, go.threshold, go.margin_hit) == \
                   (wo.start_index, wo.guard_index, wo.noise_power, wo.threshold, wo.margin_hit), (seed, cuts, edge)
        elif isinstance(g, bytes):
            assert g == w, (seed, name, cuts, edge, thr)
        else:
            assert g.tobytes() == w.tobytes(), (seed, name, cuts, edge, thr)
