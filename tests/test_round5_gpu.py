"""Round-5 additions on the GPU: the scan's tail as ONE launch of workgroup roles (noise-floor threshold, amplitude totals,
K4 screening + exact scan + record, TDOA slot: gj_capture_scan_dev / gj_part_capture_scan_dev), K5 in three launches
(start words read by the transforms, the pair's final pick by its last workgroup), and the stream-overlap probe's
failure branch.  Everything is compared with the oracle (reference arithmetic restated on the CPU) or, for bytes that
have no reference counterpart (slot layout, margins), with the separate entry points."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import gpsjam
from gpsjam import _ffi
from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc
import exact_restatement as ex

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
NOISE, WINDOW, FACTOR = 200000, 1000, 50.0
ONSET_T = np.dtype([("start", "<i8"), ("noise", "<f4"), ("thr", "<f4"), ("hit", "<f4"), ("before", "<f4"), ("guard", "<i8")])


def _levels(n, pieces, seed):
    """uint8 I/Q of n samples: Gaussian noise whose per-component sigma (LSB) is piecewise constant; pieces =
    [(first sample, sigma), ...].  Inputs are made in the test and go to the oracle and the GPU alike."""
    rng = np.random.default_rng(seed)
    sig = np.empty(n, np.float64)
    for (s0, sg), nxt in zip(pieces, pieces[1:] + [(n, 0.0)]):
        sig[s0:nxt[0]] = sg
    g = rng.standard_normal((n, 2)) * sig[:, None]
    return (np.clip(np.trunc(g), -128, 127) + 128).astype(np.uint8).reshape(-1)


def _onset(buf):
    return np.frombuffer(buf.download(np.uint8, 32).tobytes(), ONSET_T)[0]


def _fused(dev, raw, thr=0.0, chunk=65536, slice_samples=50000, want_thr=True, noise=NOISE, window=WINDOW):
    """gj_capture_scan_dev on `raw`: (power, stats, mask, amp bytes, onset record, slot bytes)."""
    n = raw.size
    buf = dev.alloc(max(n, 16) + 16).upload(raw)
    nch = dev.chunk_count(n, chunk)
    sb = dev.tdoa_slot_bytes(slice_samples)
    d_pow, d_st, d_mask, d_amp, d_on, d_slot = (dev.alloc(4 * max(nch, 1)), dev.alloc(16), dev.alloc(max(nch, 1)), dev.alloc(32),
                                                dev.alloc(32), dev.alloc(sb))
    dev.capture_scan_dev(buf, n, chunk, d_pow, thr, d_amp, noise, window, FACTOR, d_on,
                         d_stats=d_st if (want_thr and nch) else None, d_mask=d_mask if (want_thr and nch) else None,
                         slice_samples=slice_samples, d_slot=d_slot)
    dev.synchronize()
    out = (d_pow.download(np.float32, nch), d_st.download(np.float32, 3), d_mask.download(np.uint8, nch),
           d_amp.download(np.uint8, 32).tobytes(), _onset(d_on), d_slot.download(np.uint8, sb).tobytes())
    for b in (buf, d_pow, d_st, d_mask, d_amp, d_on, d_slot):
        b.free()
    return out


def _separate(dev, raw, thr=0.0, chunk=65536, slice_samples=50000, noise=NOISE, window=WINDOW):
    """The same results from the one-kernel-per-quantity entry points (K1, K3, K4 alone; threshold; slot)."""
    n = raw.size
    buf = dev.alloc(max(n, 16) + 16).upload(raw)
    nch = dev.chunk_count(n, chunk)
    sb = dev.tdoa_slot_bytes(slice_samples)
    d_pow, d_st, d_mask, d_amp, d_on, d_slot = (dev.alloc(4 * max(nch, 1)), dev.alloc(16), dev.alloc(max(nch, 1)), dev.alloc(32),
                                                dev.alloc(32), dev.alloc(sb))
    dev.chunk_power_dev(buf, n, chunk, d_pow)
    dev.amp_stats_dev(buf, n, thr, d_amp)
    dev.onset_dev(buf, n, noise, window, FACTOR, d_on)
    if nch:
        dev.power_threshold_dev(d_pow, nch, d_st, d_mask)
    dev.tdoa_slot_dev(buf, n, d_on, slice_samples, d_slot)
    dev.synchronize()
    out = (d_pow.download(np.float32, nch), d_st.download(np.float32, 3), d_mask.download(np.uint8, nch),
           d_amp.download(np.uint8, 32).tobytes(), _onset(d_on), d_slot.download(np.uint8, sb).tobytes())
    for b in (buf, d_pow, d_st, d_mask, d_amp, d_on, d_slot):
        b.free()
    return out


def _same(got, want, what, amp_exact=True):
    np.testing.assert_array_equal(got[0], want[0], err_msg=f"{what}: power map")
    if want[0].size:
        np.testing.assert_array_equal(got[1], want[1], err_msg=f"{what}: baseline / threshold / count")
        np.testing.assert_array_equal(got[2], want[2], err_msg=f"{what}: mask")
    a, b = np.frombuffer(got[3], _AMP)[0], np.frombuffer(want[3], _AMP)[0]
    assert (a["i"], a["c"]) == (b["i"], b["c"]), what
    if amp_exact:                                       # since round 6 K3 alone IS the pass: same tiles, same order, same bits
        assert got[3] == want[3], what
    else:
        np.testing.assert_allclose(a["s"], b["s"], rtol=1e-7)
    for f in ("start", "noise", "thr", "hit", "guard"):  # margin_before is a bound (gpsjam.h): checked on its own
        assert got[4][f] == want[4][f], (what, f, got[4], want[4])
    assert got[5] == want[5], f"{what}: slot bytes"


_AMP = np.dtype([("i", "<i8"), ("c", "<u8"), ("s", "<f8"), ("m", "<f4"), ("r", "<f4")])


# ----------------------------------------------------------------------------- the tail against the separate kernels and the oracle
@pytest.mark.parametrize("nbytes,chunk,thr,jam", [
    (20 * 65536 + 24691, 65536, 0.0, 220000), (20 * 65536 + 24691, 131072, 0.45, 220000), (65536 * 7, 65536, 0.1, 210000),
    (65536 * 3 + 254, 65536, 0.0, 1 << 40), (600001, 65536, 0.0, 250000), (2 * (NOISE + WINDOW), 65536, 0.0, 1 << 40),
    (2 * (NOISE + WINDOW) - 2, 65536, 0.0, 1 << 40), (400000, 1000, 0.0, 1 << 40), (2, 65536, 0.0, 0), (3, 65536, 0.0, 0),
    (41 * 65536, 65536, 0.2, 1_300_000), (40_960_000, 65536, 0.0, 9_000_000)])
def test_capture_scan_equals_the_separate_entry_points(dev, nbytes, chunk, thr, jam):
    """gj_capture_scan_dev (two launches) gives the bits of chunk power + amplitude statistics + onset + noise-floor
    threshold + TDOA slot called one after the other, and the reference's numbers where it has any: ragged ends, a chunk
    of two tiles, a capture exactly / just short of noise + window samples (the reference answers -1 there,
    triangulateTDOA.py:39), chunk sizes the fused pass does not take, two- and three-byte captures, a 10-s capture."""
    n = (nbytes + 1) // 2
    raw = generate(StreamSpec(seed=nbytes & 0xffff, jam_start=jam, jam_end=1 << 40, jam_sigma=60.0), n)[:nbytes]
    got, want = _fused(dev, raw, thr, chunk), _separate(dev, raw, thr, chunk)
    _same(got, want, f"{nbytes} bytes")
    if nbytes >= 41 * 65536:      # a slice too long for the tail's last workgroup (> 256 KiB): cut by the grid-wide copy behind it
        big = 1 << 19 if nbytes > 10_000_000 else 140_000
        _same(_fused(dev, raw, thr, chunk, slice_samples=big), _separate(dev, raw, thr, chunk, slice_samples=big), f"{nbytes} bytes, long slice")
    even = raw[:2 * (nbytes // 2)]
    # ... and what the header promises, restated with exact integers on the CPU (since round 6 the "separate" entry points
    # are the same pass: two entry points agreeing with each other proves nothing on its own)
    np.testing.assert_array_equal(got[0], ex.chunk_power(raw, chunk))
    want_on = ex.onset(raw, NOISE, WINDOW, FACTOR)
    for f in ("start", "guard", "noise", "thr", "hit"):
        assert got[4][f] == want_on[f], (f, got[4], want_on)
    want_amp, a = ex.amp_stats(raw, thr), np.frombuffer(got[3], _AMP)[0]
    assert (a["i"], a["c"]) == (want_amp["first"], want_amp["count"])
    np.testing.assert_allclose(a["s"], want_amp["sum"], rtol=2e-7)
    if nbytes <= 3_000_000:
        assert got[4]["start"] == orc.tdoa_onset(orc.tdoa_unpack(even))
        k, avg = orc.rssi_amp_stats(even, thr)
        a = np.frombuffer(got[3], _AMP)[0]
        assert a["i"] == (-1 if k is None else k)
        if k is not None:
            np.testing.assert_allclose(a["m"], avg, rtol=1e-6)
    # gj_stream_scan_dev is the same two launches without the extra roles
    buf = dev.alloc(max(nbytes, 16) + 16).upload(raw)
    nch = dev.chunk_count(nbytes, chunk)
    d_pow, d_amp, d_on = dev.alloc(4 * max(nch, 1)), dev.alloc(32), dev.alloc(32)
    dev.stream_scan_dev(buf, nbytes, chunk, d_pow, thr, d_amp, NOISE, WINDOW, FACTOR, d_on)
    dev.synchronize()
    np.testing.assert_array_equal(d_pow.download(np.float32, nch), got[0])
    assert d_amp.download(np.uint8, 32).tobytes() == got[3] and d_on.download(np.uint8, 32).tobytes() == got[4].tobytes()


def test_onset_behind_a_long_plateau_below_the_threshold(dev):
    """The case the screening cannot prove quiet: interference that sits at 0.78 of the threshold for 2.4 M samples (every
    512-sample block fails the screening bound, which covers 1536 samples for a 1000-sample window) and then rises above
    it.  Three onset workgroups look at every position of the plateau exactly; the crossing lies in the third one's
    range.  Index = the reference's; the margin in front of it is the plateau's (its largest window reaches 0.88), and positive."""
    n = 3_000_000
    s = 6.0
    plateau = s * np.sqrt(0.7 * FACTOR)      # 0.78 of the threshold once truncation has shaved the quiet floor's power
    raw = _levels(n, [(0, s), (300_000, plateau), (2_700_123, plateau * 1.35)], seed=5)
    want = orc.tdoa_onset(orc.tdoa_unpack(raw))
    assert 2_690_000 < want < 2_720_000
    got, sep = _fused(dev, raw), _separate(dev, raw)
    assert got[4]["start"] == want == sep[4]["start"]
    _same(got, sep, "plateau")
    assert 0.0 < got[4]["before"] < 0.35 and got[4]["guard"] == got[4]["start"]
    # and with no rise at all: not found, after every workgroup has looked at all of its range
    quiet = _levels(n, [(0, s), (300_000, plateau)], seed=6)
    assert orc.tdoa_onset(orc.tdoa_unpack(quiet)) == -1
    got = _fused(dev, quiet)
    assert (got[4]["start"], got[4]["guard"]) == (-1, -1) and 0.0 < got[4]["before"] < 0.35
    hdr = np.frombuffer(got[5][:16], "<i8")
    assert tuple(hdr) == (-1, -1) and not any(got[5][16:])


@pytest.mark.parametrize("edge", [512 * 2048, 2 * 512 * 2048, 512 * 2048 + 2048, 512 * 700, 512 * 2048 - 2048])
def test_onset_at_the_seams_of_the_onset_workgroups(dev, edge):
    """A burst that starts a few samples around a seam -- between two onset workgroups' ranges (multiples of 2048 blocks
    of 512 samples), between two exact-scan tiles, between two blocks: the first crossing lies up to a window in front of
    the burst, so the starts are swept until crossings fall on both sides of the seam.  Index, guard index and slot
    against the oracle / the separate kernels every time."""
    n = 2 * 512 * 2048 + 300_000
    seen = set()
    for k, ds in enumerate(range(-24, 1000, 93)):
        raw = _levels(n, [(0, 6.0), (edge + ds, 70.0)], seed=100 + k)
        want = orc.tdoa_onset(orc.tdoa_unpack(raw))
        got = _fused(dev, raw, slice_samples=4096)
        assert got[4]["start"] == want, (edge, ds)
        seen.add((want - WINDOW // 2) >= edge)
        sep = _separate(dev, raw, slice_samples=4096)
        _same(got, sep, f"seam {edge} {ds}")
    assert seen == {True, False}, "the sweep must put crossings on both sides of the seam"


def test_noise_span_that_ends_inside_a_block(dev):
    """K4's noise sum is the scan's own 512-sample block sums over the span + the span's ragged end: spans that are not
    multiples of 512 (or of 8) samples, and one shorter than a block."""
    n = 700_000
    raw = generate(StreamSpec(seed=31, jam_start=420_000, jam_end=1 << 40, jam_sigma=55.0), n)
    for noise in (200_000, 199_999, 123_457, 511, 8, 1):
        got = _fused(dev, raw, noise=noise)
        z = orc.tdoa_unpack(raw)
        assert got[4]["start"] == orc.tdoa_onset(z, noise, WINDOW, FACTOR), noise
        sep = _separate(dev, raw, noise=noise)
        _same(got, sep, f"noise span {noise}")


@pytest.mark.parametrize("window", [1, 8, 511, 512, 513, 4096, 8192])
def test_onset_windows_from_one_sample_to_the_largest(dev, window):
    """K4's window from 1 to 8192 samples (the library's limit): the exact scan's LDS tile is sized by the window
    (2048 positions + window - 1 samples), the screening's reach is (window + 510) / 512 + 1 blocks."""
    n, burst = (1_300_000, 700_123) if window <= 513 else (400_000, 250_123)      # the oracle's np.convolve is O(n * window)
    raw = _levels(n, [(0, 6.0), (burst, 60.0)], seed=40 + window)
    got = _fused(dev, raw, window=window, noise=100_000, slice_samples=4096)
    z = orc.tdoa_unpack(raw)
    assert got[4]["start"] == orc.tdoa_onset(z, 100_000, window, FACTOR), window
    _same(got, _separate(dev, raw, window=window, noise=100_000, slice_samples=4096), f"window {window}")


def test_tail_launches_back_to_back_leave_their_counter_at_zero(dev):
    """Forty fused scans of alternating captures queued without a host synchronisation in between: each launch finds the
    arrival counter at zero (the previous one's last workgroup put it back) and sees only its own records."""
    n = 1_500_000
    raws = [generate(StreamSpec(seed=70 + k, jam_start=js, jam_end=1 << 40, jam_sigma=60.0), n)
            for k, js in enumerate((400_000, 1_100_000, 1 << 40))]
    bufs = [dev.alloc(2 * n + 16).upload(r) for r in raws]
    want = [_fused(dev, r, slice_samples=8192) for r in raws]
    nch, sb = dev.chunk_count(2 * n, 65536), dev.tdoa_slot_bytes(8192)
    outs = []
    for it in range(40):
        k = it % 3
        o = (dev.alloc(4 * nch), dev.alloc(16), dev.alloc(32), dev.alloc(32), dev.alloc(sb))
        dev.capture_scan_dev(bufs[k], 2 * n, 65536, o[0], 0.0, o[2], NOISE, WINDOW, FACTOR, o[3], d_stats=o[1], slice_samples=8192,
                             d_slot=o[4])
        outs.append((k, o))
    dev.synchronize()
    for k, o in outs:
        np.testing.assert_array_equal(o[0].download(np.float32, nch), want[k][0])
        np.testing.assert_array_equal(o[1].download(np.float32, 3), want[k][1])
        assert o[2].download(np.uint8, 32).tobytes() == want[k][3]
        assert o[3].download(np.uint8, 32).tobytes() == want[k][4].tobytes()
        assert o[4].download(np.uint8, sb).tobytes() == want[k][5]
        for b in o:
            b.free()
    for b in bufs:
        b.free()


def test_part_capture_scan_cuts_the_slot_of_part_slot(dev):
    """gj_part_capture_scan_dev = gj_part_scan_dev + gj_part_slot_dev at the part's own onset: parts of a capture with
    the onset in the second part, one whose slice runs into the tail behind its own range, one that holds nothing."""
    TILE, SLICE = 65536, 30000
    n = 40 * TILE
    raw = generate(StreamSpec(seed=12, jam_start=900_000, jam_end=1 << 40, jam_sigma=60.0), n // 2)
    cuts = [0, 13 * TILE, 28 * TILE, n]
    d_noise = dev.alloc(2 * NOISE).upload(raw[:2 * NOISE])
    sb = dev.tdoa_slot_bytes(SLICE)
    for g in range(3):
        halo = TILE if g else 0
        b0, b1 = cuts[g] - halo, min(n, cuts[g + 1] + 2 * SLICE + TILE)
        buf = dev.alloc(b1 - b0 + 16).upload(raw[b0:b1])
        view = _ffi.PartView(buf.ptr, b1 - b0, b0, cuts[g], cuts[g + 1] - cuts[g], n, d_noise.ptr)
        nch, nt = (cuts[g + 1] - cuts[g]) // TILE, (cuts[g + 1] - cuts[g]) // TILE
        res = []
        for fused in (True, False):
            d_pow, d_tiles, d_amp, d_on, d_slot = dev.alloc(4 * nch), dev.alloc(16 * nt), dev.alloc(32), dev.alloc(32), dev.alloc(sb)
            if fused:
                dev.part_capture_scan_dev(view, 65536, d_pow, 0.0, d_tiles, d_amp, NOISE, WINDOW, FACTOR, d_on, slice_samples=SLICE,
                                          d_slot=d_slot)
            else:
                dev.part_scan_dev(view, 65536, d_pow, 0.0, d_tiles, d_amp, NOISE, WINDOW, FACTOR, d_on)
                dev.part_slot_dev(view, d_on, SLICE, d_slot)
            dev.synchronize()
            res.append(tuple(b.download(np.uint8).tobytes() for b in (d_pow, d_tiles, d_amp, d_on, d_slot)))
            for b in (d_pow, d_tiles, d_amp, d_on, d_slot):
                b.free()
        assert res[0] == res[1], f"part {g}"
        on = np.frombuffer(res[0][3], ONSET_T)[0]
        want = orc.tdoa_onset(orc.tdoa_unpack(raw))
        assert on["start"] == (want if cuts[g] // 2 <= want - WINDOW // 2 < cuts[g + 1] // 2 else on["start"])
        buf.free()
    d_noise.free()


# ----------------------------------------------------------------------------- K5 in three launches
def test_k5_repeated_solves_pick_the_same_winner(dev, g4_raws, golden_meta):
    """The pair's final pick is made by whichever of its workgroups finishes last: thirty solves queued back to back
    (per-pair arrival counters back at zero each time) give the golden lags, and bit-identical peaks and margins."""
    sl = 50000
    on = [orc.tdoa_onset(orc.tdoa_unpack(r)) for r in g4_raws]
    assert on == golden_meta["g4"]["onset"]
    sb = dev.tdoa_slot_bytes(sl)
    slots = dev.alloc(3 * sb)
    caps = [dev.alloc(r.size + 16).upload(r) for r in g4_raws]
    d_on = dev.alloc(32 * 3)
    for a in range(3):
        dev.onset_dev(caps[a], g4_raws[a].size, NOISE, WINDOW, FACTOR, d_on.ptr + 32 * a)
        dev.tdoa_slot_dev(caps[a], g4_raws[a].size, d_on.ptr + 32 * a, sl, slots.ptr + a * sb)
    pairs = [(0, 1), (0, 2), (1, 2)]
    outs = []
    for _ in range(30):
        o = (dev.alloc(12), dev.alloc(12), dev.alloc(12))
        dev.xcorr_slots_dev(slots, sb, 3, sl, pairs, o[0], o[1], o[2])
        outs.append(o)
    dev.synchronize()
    own = golden_meta["g4"]["lags_own_start"]
    want = [own["50000_01"], own["50000_02"], own["50000_12"]]
    first = None
    for o in outs:
        got = (o[0].download(np.int32, 3), o[1].download(np.float32, 3), o[2].download(np.float32, 3))
        assert list(got[0]) == want
        if first is None:
            first = got
        assert all(np.array_equal(a, b) for a, b in zip(got, first))
        for b in o:
            b.free()
    # an antenna whose slot is invalid: its pairs answer INVALID, the other pair is untouched
    bad = np.frombuffer(slots.download(np.uint8, 3 * sb).tobytes(), np.uint8).copy()
    bad[2 * sb:2 * sb + 8] = 255
    slots.upload(bad)
    o = (dev.alloc(12), dev.alloc(12), dev.alloc(12))
    dev.xcorr_slots_dev(slots, sb, 3, sl, pairs, o[0], o[1], o[2])
    dev.synchronize()
    lags = o[0].download(np.int32, 3)
    assert lags[0] == want[0] and lags[1] == lags[2] == -(1 << 31)
    assert list(o[1].download(np.float32, 3)[1:]) == [0.0, 0.0]


# ----------------------------------------------------------------------------- the stream-overlap probe's failure branch
CHILD = r"""
import hashlib, logging, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "gps-jamming_amd"))
import torch
import gpsjam
from gpsjam import local
from gpsjam.synth import StreamSpec, generate
logging.basicConfig(stream=sys.stdout, level=logging.WARNING, format="LOG %(name)s %(message)s")
n = 900_000
raws = [generate(StreamSpec(seed=61, antenna=a, delay=d, jam_start=400_000, jam_end=800_000, jam_sigma=s), n)
        for a, (d, s) in enumerate(((0, 60.0), (7, 52.0), (-4, 56.0)))]
dev = gpsjam.Device(0)
work = torch.cuda.Stream()
torch.cuda.set_stream(work)
dev.set_stream(work.cuda_stream)
caps = [torch.from_numpy(r).cuda() for r in raws]
with local.LocalAntennas(dev, caps, chunk_samples=131072, nperseg=1024, slice_samples=50000, graph=False) as st:
    for _ in range(3):
        got = st.step()
    got.wait()
    torch.cuda.synchronize()
    print("OVERLAP", st.streams_overlap)
    print("RESULT", hashlib.sha256(got.vectors.cpu().numpy().tobytes()).hexdigest())
    res, td = got.unpack()
    print("LAGS", td.lags, [r.onset for r in res])
"""


def test_pipelines_stay_correct_when_no_stream_overlaps(tmp_path):
    """GPU_MAX_HW_QUEUES=1: the runtime maps every stream onto ONE hardware queue, so no candidate can run beside the
    main stream.  stream_beside gives up after its tries with a warning; the deployment step -- whose cross-stream order
    is by events, not by luck -- still gives byte for byte the results of a process with the default four queues."""
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    outs = {}
    for queues in ("1", None):
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        if queues:
            env["GPU_MAX_HW_QUEUES"] = queues
        p = subprocess.run([sys.executable, str(script), REPO], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        outs[queues] = p.stdout
    one, four = outs["1"], outs[None]
    assert "no stream found that runs beside" in one, one
    assert "no stream found that runs beside" not in four, four
    pick = lambda text, key: [ln for ln in text.splitlines() if ln.startswith(key)]
    assert pick(one, "RESULT") == pick(four, "RESULT") and len(pick(one, "RESULT")) == 1
    assert pick(one, "LAGS") == pick(four, "LAGS") and len(pick(one, "LAGS")) == 1
    assert pick(one, "OVERLAP") == ["OVERLAP False"] and pick(four, "OVERLAP") == ["OVERLAP True"]   # and the pipeline says which it is


def test_probe_cap_and_checked_search(dev):
    """gj_probe_busy_dev refuses more than 100 ms; stream_beside_checked says whether its stream overlaps."""
    import torch
    from gpsjam import streams
    with pytest.raises(gpsjam.GpsJamError):
        dev.probe_busy_dev(100.5)
    main = torch.cuda.Stream()
    dev.set_stream(main.cuda_stream)
    try:
        s, ok = streams.stream_beside_checked([(dev, main)])
        assert ok and streams.runs_beside(dev, main, s) and not streams.runs_beside(dev, main, main)
    finally:
        dev.set_stream(None, external=False)


# ----------------------------------------------------------------------------- K2's wave-fence exchange against plain barriers
WELCH_CHILD = r"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "gps-jamming_amd"))
import numpy as np
import gpsjam
from gpsjam.synth import StreamSpec, generate
raw = generate(StreamSpec(seed=404, jam_start=300_000, jam_end=900_000, jam_sigma=45.0, dc_i_q8=384, dc_q_q8=-192), 1_200_000)
with gpsjam.Device(0) as dev:
    print("LIB", os.path.basename(gpsjam.library_path()))
    for n in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        for chunk in (200_000, 131_072 + 7 * n):
            psd, _ = dev.welch(raw, chunk_samples=chunk, nperseg=n, want_db=False)
            print("PSD", n, chunk, psd.shape[0], hashlib.sha256(np.ascontiguousarray(psd).tobytes()).hexdigest())
"""


def test_welch_wave_fence_exchange_equals_the_barrier_build(tmp_path):
    """For N <= 1024 K2 orders its LDS exchange with a wavefront fence instead of a workgroup barrier (a transform lies
    inside one wave).  The same sources built with -DGJ_W_WAVEFENCE=0 (csrc/libgpsjam_hip_barrier.so, made by the
    Makefile next to the product library) must give the same PSD bytes at every size, full and ragged chunks -- a race
    the removed barriers used to hide would show as differing bits."""
    libdir = os.path.join(REPO, "gps-jamming_amd", "csrc")
    variant = os.path.join(libdir, "libgpsjam_hip_barrier.so")
    if not os.path.exists(variant):
        subprocess.run(["make", "-C", libdir, "-j", "8", "libgpsjam_hip_barrier.so"], check=True, timeout=900)
    script = tmp_path / "welch_child.py"
    script.write_text(WELCH_CHILD)
    outs = {}
    for name, lib in (("fence", None), ("barrier", variant)):
        env = dict(os.environ)
        env.pop("GPSJAM_LIB", None)
        if lib:
            env["GPSJAM_LIB"] = lib
        p = subprocess.run([sys.executable, str(script), REPO], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        outs[name] = p.stdout.splitlines()
    assert outs["fence"][0] == "LIB libgpsjam_hip.so" and outs["barrier"][0] == "LIB libgpsjam_hip_barrier.so"
    assert len(outs["fence"]) == 19 and outs["fence"][1:] == outs["barrier"][1:]


# ----------------------------------------------------------------------------- one K2 launch / one pack launch for a deployment
@pytest.mark.parametrize("nperseg,chunk,n", [(1024, 131072, 900_000), (4096, 200_000, 1_000_000), (64, 65536, 300_001), (1024, 2048000, 20_480_000)])
def test_welch_batch_gives_each_capture_the_bits_of_its_own_launch(dev, nperseg, chunk, n):
    """gj_welch_batch_dev: three captures of one length in one transform launch + one finalize; every PSD byte equal to
    gj_welch_dev on the capture alone (same plan, same partial spectra, same fixed order of summation) -- full chunks, a
    ragged last chunk, the reference's 10-s shape."""
    raws = [generate(StreamSpec(seed=500 + a, antenna=a, jam_start=n // 3, jam_end=2 * n // 3, jam_sigma=30.0 + 10 * a), n) for a in range(3)]
    caps = [dev.alloc(2 * n + 16).upload(r) for r in raws]
    rows = dev.welch_rows(2 * n, chunk, nperseg)
    assert rows >= 1
    alone = [dev.alloc(4 * rows * nperseg) for _ in range(3)]
    batch = [dev.alloc(4 * rows * nperseg) for _ in range(3)]
    for a in range(3):
        dev.welch_dev(caps[a], 2 * n, chunk, nperseg, 2.048e6, alone[a])
    dev.welch_batch_dev(caps, 2 * n, chunk, nperseg, 2.048e6, batch)
    dev.synchronize()
    for a in range(3):
        assert alone[a].download(np.uint8).tobytes() == batch[a].download(np.uint8).tobytes(), f"capture {a}"
    lin, _, _ = orc.widmo_waterfall(raws[1][:2 * min(n, 1_000_000)], nperseg=nperseg, chunk_samples=chunk) if n <= 1_000_000 else (None, None, None)
    if lin is not None:
        got = batch[1].download(np.float32, rows * nperseg).reshape(rows, nperseg)
        keep = lin > 1e-12
        assert float(np.max(np.abs(got[keep] - lin[keep]) / lin[keep])) < 1e-4
    for b in caps + alone + batch:
        b.free()
    with pytest.raises(gpsjam.GpsJamError):
        dev.welch_batch_dev([], 2 * n, chunk, nperseg, 2.048e6, [])


def test_pack_results_in_one_launch_equals_one_launch_per_capture(dev):
    """gj_pack_results_dev against gj_pack_result_dev, byte for byte: three captures, antenna 0 carrying the pair table."""
    nperseg, n_ant = 1024, 3
    rng = np.random.default_rng(9)
    nch, rows = [625, 625, 400], [10, 10, 7]
    f32 = lambda k: rng.random(k).astype(np.float32)
    bufs = []

    def up(a):
        b = dev.alloc(max(a.nbytes, 16)).upload(a.view(np.uint8))
        bufs.append(b)
        return b
    power = [up(f32(k)) for k in nch]
    stats = [up(f32(3)) for _ in range(n_ant)]
    amp = [up(rng.integers(0, 1 << 40, 4).astype(np.int64)) for _ in range(n_ant)]
    onset = [up(rng.integers(0, 1 << 30, 4).astype(np.int64)) for _ in range(n_ant)]
    psd = [up(f32(r * nperseg)) for r in rows]
    pairs, lags = up(np.array([0, 1, 0, 2, 1, 2], np.int32)), up(np.array([3, -5, -8], np.int32))
    peaks, margins = up(f32(3)), up(f32(3))
    from gpsjam.sharded import result_len
    ln = max(result_len(k, nperseg, 3) for k in nch)
    one = [dev.alloc(8 * ln) for _ in range(n_ant)]
    many = [dev.alloc(8 * ln) for _ in range(n_ant)]
    for b in one + many:
        b.upload(np.zeros(8 * ln, np.uint8))
    desc = []
    for a in range(n_ant):
        carries = a == 0
        dev.pack_result_dev(nch[a], power[a], stats[a], amp[a], onset[a], psd[a], rows[a], nperseg, a, 3 if carries else 0, 3,
                            pairs if carries else None, lags if carries else None, peaks if carries else None,
                            margins if carries else None, one[a])
        desc.append(_ffi.CombineCapture(n_chunks=nch[a], rows=rows[a], n_tiles=0, total_bytes=0, n_parts=1, antenna=a,
                                        n_pairs=3 if carries else 0, pair_cap=3, d_power=power[a].ptr, d_stats=stats[a].ptr,
                                        d_tiles=None, d_amp_parts=None, d_onset_parts=None, d_amp=amp[a].ptr, d_onset=onset[a].ptr,
                                        d_psd=psd[a].ptr, d_out=many[a].ptr))
    dev.pack_results_dev(desc, nperseg, pairs, lags, peaks, margins)
    dev.synchronize()
    for a in range(n_ant):
        assert one[a].download(np.uint8).tobytes() == many[a].download(np.uint8).tobytes(), f"capture {a}"
    for b in bufs + one + many:
        b.free()


# ----------------------------------------------------------------------------- fill threads of a staged copy
def test_fill_threads_change_the_copy_not_the_results(dev):
    """gj_set_fill_threads: 0 = by capture size (four for a 10-s capture), 1..16 fixed.  Pieces and threads differ, the
    bytes in HBM and every result that rides on the capture do not; values outside 0..16 are refused."""
    raw = generate(StreamSpec(seed=77, jam_start=3_000_000, jam_end=1 << 40, jam_sigma=50.0), 9_000_000)    # 18 MB: staged
    want = None
    try:
        for n in (0, 1, 2, 5, 16):
            dev.set_fill_threads(n)
            with dev.ingest(raw, rssi_threshold=0.0, welch=(2048000, 1024)) as cap:
                got = (hashlib.sha256(cap.download().tobytes()).hexdigest(), dev.chunk_power(cap).tobytes(), bytes(dev.amp_stats(cap, 0.0)),
                       bytes(dev.onset(cap)), dev.welch(cap, nperseg=1024, want_db=False)[0].tobytes())
            want = want or got
            assert got == want, n
            with dev.capture(raw) as cap:
                assert hashlib.sha256(cap.download().tobytes()).hexdigest() == want[0]
        for bad in (-1, 17):
            with pytest.raises(gpsjam.GpsJamError):
                dev.set_fill_threads(bad)
    finally:
        dev.set_fill_threads(0)
    assert want[0] == hashlib.sha256(raw.tobytes()).hexdigest()


# ----------------------------------------------------------------------------- packing on the second stream (optional)
def test_antenna_stream_with_the_packing_on_the_second_stream():
    """AntennaStream(pack_on_side=True): the main stream carries K2 + finalize only, the result vector is packed behind K5 on
    the second stream from one of two PSD buffers.  Six steps back to back: every step's vector byte-equal to the default
    arrangement's (both buffer sets, both PSD buffers in use)."""
    import torch
    from gpsjam import sharded
    n = 900_000
    raws = [generate(StreamSpec(seed=31, antenna=a, delay=d, jam_start=400_000, jam_end=800_000, jam_sigma=55.0), n)
            for a, d in enumerate((0, 6, -4))]
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    out = {}
    try:
        for side in (False, True):
            with gpsjam.Device(0) as dev:
                dev.set_stream(work.cuda_stream)
                caps = [torch.from_numpy(r).cuda() for r in raws]
                sl = 50000
                aux = torch.zeros((2, dev.tdoa_slot_bytes(sl)), dtype=torch.uint8, device="cuda")
                d_on = torch.zeros(4, dtype=torch.int64, device="cuda")
                for a in (1, 2):
                    dev.onset_dev(caps[a], caps[a].numel(), 200000, 1000, 50.0, d_on)
                    dev.tdoa_slot_dev(caps[a], caps[a].numel(), d_on, sl, aux[a - 1])
                st = sharded.AntennaStream(dev, caps[0], nperseg=1024, chunk_samples=131072, slice_samples=sl, aux_slots=aux,
                                           pack_on_side=side)
                assert st._pack_on_side is side and len(st.psd2) == (2 if side else 1)
                vecs = []
                for _ in range(6):
                    got = st.step()
                    got.wait()
                    vecs.append(got.vectors.clone())
                torch.cuda.synchronize()
                res, td = got.unpack()
                out[side] = ([v.cpu().numpy().tobytes() for v in vecs], td.lags, st.psd[:st.rows].cpu().numpy().tobytes())
                st.close()
        assert out[True] == out[False] and len(set(out[True][0])) == 1
        assert len(out[True][1]) == 3 and all(abs(l) < 64 for l in out[True][1])      # three solved pairs, slots cut at their own onsets
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())


# ----------------------------------------------------------------------------- a deployment's files brought in together
def test_ingest_many_equals_ingest_file_by_file(dev, tmp_path):
    """gj_ingest_files (Device.ingest_many): the recordings of a deployment uploaded and analysed side by side by threads of
    the library's own -- resident bytes and every ride-along result equal to Device.ingest on each file; files of unequal
    length; a file the library cannot open reports which one, and nothing stays allocated behind it."""
    ns = (2_000_000, 2_000_000, 1_234_567, 300)
    paths = []
    for a, n in enumerate(ns):
        raw = generate(StreamSpec(seed=880 + a, antenna=a % 3, jam_start=min(900_000, n // 2), jam_end=1 << 40, jam_sigma=40.0 + 8 * a), n)
        p = tmp_path / f"rec{a}.bin"
        raw.tofile(p)
        paths.append(str(p))
    kw = dict(rssi_threshold=0.1, welch=(131072, 1024), want_db=True)

    def facts(c):
        return (hashlib.sha256(c.download().tobytes()).hexdigest(), dev.chunk_power(c).tobytes(), bytes(dev.amp_stats(c, 0.1)),
                bytes(dev.onset(c)), tuple(a.tobytes() for a in dev.welch(c, chunk_samples=131072, nperseg=1024)))
    one = [dev.ingest(p, **kw) for p in paths]
    want = [facts(c) for c in one]
    for c in one:
        c.free()
    for rounds in range(3):
        hits = dev.cache_hits
        many = dev.ingest_many(paths, **kw)
        got = [facts(c) for c in many]
        assert dev.cache_hits >= hits + 4 * 3                      # served from the captures: the results rode in with the uploads
        assert got == want
        assert [c.nbytes for c in many] == [2 * n for n in ns] and len({c.ptr for c in many}) == 4
        for c in many:
            c.free()
    assert dev.ingest_many([]) == []
    with pytest.raises(FileNotFoundError):
        dev.ingest_many([paths[0], str(tmp_path / "nothing.bin")], **kw)
    os.chmod(paths[2], 0)
    try:
        if os.geteuid() != 0:                                      # root reads anything: the library-side failure needs a plain user
            with pytest.raises(gpsjam.GpsJamError) as e:
                dev.ingest_many(paths, **kw)
            assert "file 2 of 4" in str(e.value)
    finally:
        os.chmod(paths[2], 0o644)
    c = dev.ingest_many(paths[:1], **kw)                            # one job: no thread is started
    assert facts(c[0]) == want[0]
    c[0].free()
    assert dev.debug_counters()["lanes_busy"] == 0

