"""Seeded random sweep of the five kernels against the CPU oracle: sizes, chunkings, thresholds,
onset positions and delays nobody hand-picked (the fixed cases live in test_gpu_parity.py).
Same tolerances as there: power map 1e-6 rel, PSD 1e-4 rel on the linear PSD, amplitude mean 1e-6 rel, onset index and lag bit-exact."""
import numpy as np
import pytest

from gpsjam.synth import StreamSpec, generate
from oracle import gpsjam_oracle as orc

pytestmark = pytest.mark.gpu

CASES = list(range(12))


def _rng(case, salt):
    return np.random.RandomState(1000 * salt + case)


def rel_err(got, want, floor=1e-12):
    keep = want > floor
    return float(np.max(np.abs(got[keep] - want[keep]) / want[keep]))


@pytest.mark.parametrize("case", CASES)
def test_sweep_k1_power_and_threshold(dev, case):
    r = _rng(case, 1)
    chunk_bytes = int(r.choice([2, 6, 1000, 4096, 65536, 131072, 200000]))
    nbytes = int(r.randint(1, 40)) * chunk_bytes + int(r.randint(0, chunk_bytes))
    nbytes = max(nbytes, 2)
    nsamp = (nbytes + 1) // 2
    spec = StreamSpec(seed=500 + case, jam_start=int(nsamp * r.uniform(0.2, 0.6)), jam_end=int(nsamp * r.uniform(0.6, 0.95)),
                      jam_sigma=float(r.uniform(15, 70)), dc_i_q8=int(r.randint(-2000, 2000)))
    raw = generate(spec, nsamp)[:nbytes]
    got = dev.chunk_power(raw, chunk_bytes=chunk_bytes)
    want = orc.chunk_power(raw, chunk_bytes=chunk_bytes)
    assert got.shape == want.shape
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6)
    if fin.all() and got.size >= 1:
        d_pow, d_stats, d_mask = dev.alloc(4 * got.size).upload(got), dev.alloc(12), dev.alloc(got.size)
        dev.power_threshold_dev(d_pow, got.size, d_stats, d_mask)
        dev.synchronize()
        base, thr, _ = orc.power_threshold(got)
        stats = d_stats.download(np.float32)
        assert stats[0] == np.float32(base)
        np.testing.assert_allclose(stats[1], np.float32(thr), rtol=2e-7)


@pytest.mark.parametrize("case", CASES)
def test_sweep_k2_welch(dev, case):
    r = _rng(case, 2)
    nperseg = int(r.choice([16, 64, 256, 1024, 2048, 4096]))
    chunk = int(r.randint(2 * nperseg, 6 * nperseg + 3000))
    nchunks = int(r.randint(1, 4))
    tail = int(r.randint(0, chunk))
    nsamp = nchunks * chunk + tail
    spec = StreamSpec(seed=700 + case, jam_start=int(nsamp * 0.3), jam_end=int(nsamp * 0.8), jam_sigma=float(r.uniform(10, 60)),
                      dc_i_q8=int(r.randint(-3000, 3000)), dc_q_q8=int(r.randint(-3000, 3000)))
    raw = generate(spec, nsamp)
    psd, db = dev.welch(raw, chunk_samples=chunk, nperseg=nperseg)
    lin, dbo, _ = orc.widmo_waterfall(raw, nperseg=nperseg, chunk_samples=chunk)
    assert psd.shape == lin.shape
    if lin.size:
        assert rel_err(psd, lin) < 1e-4
        np.testing.assert_allclose(db, dbo, atol=5e-4)


@pytest.mark.parametrize("case", CASES)
def test_sweep_k3_amp_stats(dev, case):
    r = _rng(case, 3)
    nsamp = int(r.randint(1, 400000))
    odd = int(r.randint(0, 2))
    thr = float(r.choice([0.0, 0.004, 0.1, 0.3, 0.9, 1.5]))
    spec = StreamSpec(seed=900 + case, jam_start=int(nsamp * r.uniform(0.1, 0.9)), jam_end=1 << 40,
                      jam_sigma=float(r.uniform(20, 90)))
    raw = generate(spec, nsamp + 1)[:2 * nsamp + odd]
    st = dev.amp_stats(raw, thr)
    k, avg = orc.rssi_amp_stats(raw[:2 * nsamp], thr)
    if k is None:
        assert st.first_index == -1 and st.count == 0
    else:
        assert st.first_index == k and st.count == nsamp - k
        np.testing.assert_allclose(st.mean, avg, rtol=1e-6)


@pytest.mark.parametrize("case", CASES)
def test_sweep_k4_onset(dev, case):
    r = _rng(case, 4)
    noise = int(r.choice([1000, 8192, 50000, 200000]))
    window = int(r.choice([1, 7, 100, 1000, 4096]))
    nsamp = noise + window + int(r.randint(0, 150000))
    onset_at = int(r.randint(noise, nsamp + 20000))          # sometimes beyond the end: no onset
    factor = float(r.choice([5.0, 50.0]))
    raw = generate(StreamSpec(seed=1100 + case, jam_start=onset_at, jam_end=1 << 40, jam_sigma=float(r.uniform(40, 90))), nsamp)
    z = orc.tdoa_unpack(raw)
    assert dev.onset(raw, noise, window, factor).start_index == orc.tdoa_onset(z, noise, window, factor)


@pytest.mark.parametrize("case", CASES)
def test_sweep_k5_xcorr(dev, case):
    r = _rng(case, 5)
    n = int(r.randint(64, 90000))
    delays = [0, int(r.randint(-40, 41)), int(r.randint(-40, 41))]
    raws = [generate(StreamSpec(seed=1300 + case, antenna=a, delay=d, jam_start=-(1 << 40), jam_end=1 << 40,
                                jam_sigma=float(r.uniform(30, 70))), n) for a, d in enumerate(delays)]
    pairs = [(0, 1), (0, 2), (1, 2), (2, 0)]
    lags, peaks = dev.xcorr_lags(raws, pairs)
    zs = [orc.tdoa_unpack(x) for x in raws]
    for (a, b), lag, pk in zip(pairs, lags, peaks):
        want, wpk = orc.xcorr_lag(zs[b], zs[a])
        assert lag == want == delays[b] - delays[a]
        np.testing.assert_allclose(pk, wpk, rtol=1e-4)


@pytest.mark.parametrize("case", CASES[:8])
def test_sweep_slots_and_pair_dealing(dev, case):
    """Round 2: TDOA slots cut on the device at random onsets, the antenna pairs dealt over 'ranks' as the
    sharded pipeline deals them (sharded.pairs_of_rank), one gj_xcorr_slots_dev call per rank; the union must be
    every pair exactly once with the oracle's lag, and an invalid slot must invalidate exactly its pairs."""
    from gpsjam import sharded
    r = _rng(case, 6)
    n_ant = int(r.randint(2, 7))
    sl = int(r.choice([4096, 12345, 50000, 65536]))
    n = 260000 + sl + 5000
    delays = [0] + [int(r.randint(-30, 31)) for _ in range(n_ant - 1)]
    start = int(r.randint(215000, 250000))                   # the common burst, seen by antenna a `delays[a]` samples late
    raws = [generate(StreamSpec(seed=1500 + case, antenna=a, delay=d, jam_start=start, jam_end=1 << 40,
                                jam_sigma=float(60 + 3 * a)), n) for a, d in enumerate(delays)]
    bad = int(r.randint(0, n_ant)) if case % 3 == 0 else -1
    sb = dev.tdoa_slot_bytes(sl)
    slots, d_on = dev.alloc(n_ant * sb), dev.alloc(32)
    onsets = []
    for a, raw in enumerate(raws):
        with dev.capture(raw) as cap:
            dev.onset_dev(cap, cap.nbytes, 200000, 1000, 50.0, d_on)
            if a == bad:                                     # a slice that runs off the end of the capture
                d_on.upload(np.array([n - sl + 1], np.int64))
            dev.tdoa_slot_dev(cap, cap.nbytes, d_on, sl, slots.ptr + a * sb)
            dev.synchronize()
        onsets.append(orc.tdoa_onset(orc.tdoa_unpack(raw)))
    z = [orc.tdoa_unpack(x) for x in raws]
    table = {}
    d_l, d_p, d_m = dev.alloc(64), dev.alloc(64), dev.alloc(64)
    for rank in range(n_ant):
        mine = sharded.pairs_of_rank(rank, n_ant)
        if not mine:
            continue
        dev.xcorr_slots_dev(slots, sb, n_ant, sl, mine, d_l, d_p, d_m)
        dev.synchronize()
        for (i, j), lag in zip(mine, d_l.download(np.int32, len(mine)).tolist()):
            ci, cj, cl = sharded.canonical_pair(i, j, lag)
            assert (ci, cj) not in table
            table[(ci, cj)] = cl
    assert sorted(table) == sharded.all_pairs(n_ant)
    for (i, j), lag in table.items():
        if bad in (i, j):
            assert lag == sharded.LAG_INVALID
        else:
            want = orc.xcorr_lag(z[j][onsets[j]:onsets[j] + sl], z[i][onsets[i]:onsets[i] + sl])[0]
            assert lag == want and lag + onsets[j] - onsets[i] == delays[j] - delays[i]


@pytest.mark.parametrize("case", CASES[:8])
def test_sweep_unpack_convention(dev, case):
    """gj_set_unpack with random half-integer offsets and scales: K1 and K3 against plain numpy."""
    r = _rng(case, 7)
    offset = float(r.randint(200, 312)) / 2.0                # 100.0 .. 155.5 in steps of 0.5
    scale = float(r.choice([1.0, 1 / 127.5, 1 / 128.0, 0.01]))
    nsamp = int(r.randint(5000, 300000))
    raw = generate(StreamSpec(seed=1700 + case, jam_start=nsamp // 3, jam_end=1 << 40, jam_sigma=float(r.uniform(20, 80))), nsamp)
    v = raw.astype(np.float64) - offset
    try:
        dev.set_unpack(offset, scale)
        pm = dev.chunk_power(raw, chunk_bytes=65536, eps=0.0)
        want = [np.float32((v[o:o + 65536] ** 2).sum() / (len(v[o:o + 65536]) // 2)) for o in range(0, raw.size, 65536)]
        np.testing.assert_array_equal(pm, np.array(want, np.float32))
        amp = np.sqrt(v[0::2] ** 2 + v[1::2] ** 2) * scale
        thr = float(r.choice([0.0, 0.3])) * scale * 127.5
        st = dev.amp_stats(raw, thr)
        hits = np.nonzero(amp.astype(np.float32) > np.float32(thr))[0]
        if hits.size:
            k = int(hits[0])
            assert abs(st.first_index - k) <= 0 or abs(amp[st.first_index] - thr) < 1e-6 * max(thr, 1e-9)
            np.testing.assert_allclose(st.mean, amp[st.first_index:].mean(), rtol=2e-6)
        else:
            assert st.first_index == -1
    finally:
        dev.set_unpack()
