"""The per-GPU pipeline object used by bench.py (gpsjam.sharded.AntennaStream) on one GPU:
device-side result packing against the torch reference packer, and the unpacked results
against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_antenna_stream_single_gpu():
    import torch
    import gpsjam
    from gpsjam import sharded
    from gpsjam.synth import StreamSpec, generate
    from oracle import gpsjam_oracle as orc

    n = 700000
    spec = StreamSpec(seed=21, antenna=0, delay=0, jam_start=300000, jam_end=1 << 40, jam_sigma=60.0)
    raw = generate(spec, n)
    dev = gpsjam.Device(0)
    stream_handle = torch.cuda.Stream()
    torch.cuda.set_stream(stream_handle)
    dev.set_stream(stream_handle.cuda_stream)
    cap = torch.from_numpy(raw).cuda()
    st = sharded.AntennaStream(dev, cap, nperseg=1024, chunk_samples=200000, slice_samples=65536)
    got = st.step()
    torch.cuda.synchronize()
    assert len(got) == 1
    # the kernel-packed vector equals the torch-packed one
    ref = sharded.pack_stream_reference(st)
    np.testing.assert_allclose(got[0].cpu().numpy(), ref.cpu().numpy(), rtol=1e-6)
    np.testing.assert_array_equal(got[0][:sharded.HEADER + st.n_chunks].cpu().numpy(),
                                  ref[:sharded.HEADER + st.n_chunks].cpu().numpy())
    res = sharded.unpack_results(got[0])
    pm = orc.chunk_power(raw)
    np.testing.assert_allclose(res.power_map, pm, rtol=1e-6)
    base, thr, ranges = orc.power_threshold(pm)
    assert res.baseline == np.float32(base) and res.jamming_byte_ranges() == [(int(a), int(b)) for a, b in ranges]
    k, avg = orc.rssi_amp_stats(raw, 0.0)
    assert res.amp_first == k and res.amp_count == n - k
    np.testing.assert_allclose(res.amp_mean, avg, rtol=1e-6)
    z = orc.tdoa_unpack(raw)
    assert res.onset == orc.tdoa_onset(z)
    assert res.onset_guard == res.onset and res.onset_margin_hit > 1e-6 and not res.onset_near_tie
    assert res.onset_threshold == np.float32(res.noise_power) * np.float32(50.0)
    assert res.lag == 0                                   # rank 0 is the reference antenna
    results, td = got.unpack()
    assert td.pairs == [] and results[0].onset == res.onset   # one antenna: nothing to correlate
    lin, _, _ = orc.widmo_waterfall(raw, nperseg=1024, chunk_samples=200000)
    np.testing.assert_allclose(res.mean_spectrum, lin.mean(axis=0), rtol=1e-4)
    torch.cuda.set_stream(torch.cuda.default_stream())
    dev.close()


@pytest.mark.parametrize("own_stream", [True, False])
def test_antenna_stream_overlap_matches_single_stream(own_stream):
    """Two-stream pipeline (scan / threshold / TDOA beside K2) against the single-stream order,
    several steps back to back: the result vector must not depend on the overlap and must not
    change from step to step (the cross-step events keep pack and the next scan apart)."""
    import torch
    import gpsjam
    from gpsjam import sharded
    from gpsjam.synth import StreamSpec, generate

    n = 3_000_000
    raw = generate(StreamSpec(seed=5, antenna=1, delay=2, jam_start=1_200_000, jam_end=2_100_000, jam_sigma=50.0), n)
    dev = gpsjam.Device(0)
    if own_stream:
        work = torch.cuda.Stream()
        torch.cuda.set_stream(work)
        dev.set_stream(work.cuda_stream)
    else:                                   # torch's default stream, context never told about it
        torch.cuda.set_stream(torch.cuda.default_stream())
    cap = torch.from_numpy(raw).cuda()
    vecs = {}
    for overlap in (False, True):
        st = sharded.AntennaStream(dev, cap, nperseg=4096, chunk_samples=500000, slice_samples=1 << 16, overlap=overlap)
        assert st.overlap is overlap
        outs = []
        for _ in range(4):
            outs.append(st.step()[0].clone())
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(o, outs[0])
        vecs[overlap] = outs[0].cpu().numpy()
    np.testing.assert_array_equal(vecs[True], vecs[False])
    res = sharded.unpack_results(torch.from_numpy(vecs[True]))
    assert res.onset > 0 and res.lag == 0 and len(res.jamming_byte_ranges()) == 1
    torch.cuda.set_stream(torch.cuda.default_stream())
    dev.close()


@pytest.mark.parametrize("overlap", [True, False])
def test_antenna_stream_three_antennas_one_gpu(overlap):
    """BASELINE configs[3] inside the per-stream pipeline: this capture's TDOA slot + the slots of
    two further antennas (cut by gj_tdoa_slot_dev at their own onsets), all three pairs solved by
    one gj_xcorr_slots_dev launch.  Lags against the oracle on the same slices and against the
    delays the captures were built with."""
    import torch
    import gpsjam
    from gpsjam import sharded
    from gpsjam.synth import StreamSpec, generate
    from oracle import gpsjam_oracle as orc

    n, sl = 900000, 1 << 16
    delays = (0, 4, -7)
    raws = [generate(StreamSpec(seed=31, antenna=a, delay=d, jam_start=400000, jam_end=1 << 40, jam_sigma=55.0), n)
            for a, d in enumerate(delays)]
    dev = gpsjam.Device(0)
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    dev.set_stream(work.cuda_stream)
    caps = [torch.from_numpy(r).cuda() for r in raws]
    sb = dev.tdoa_slot_bytes(sl)
    assert sb == sharded.slot_bytes(sl)
    aux = torch.zeros((2, sb), dtype=torch.uint8, device="cuda")
    d_on = torch.zeros(4, dtype=torch.int64, device="cuda")
    onsets = [orc.tdoa_onset(orc.tdoa_unpack(r)) for r in raws]
    for a in (1, 2):
        dev.onset_dev(caps[a], caps[a].numel(), 200000, 1000, 50.0, d_on)
        dev.tdoa_slot_dev(caps[a], caps[a].numel(), d_on, sl, aux[a - 1])
        torch.cuda.synchronize()
        assert int(d_on[0]) == onsets[a]
        # the kernel-made slot equals the torch-made one
        want = sharded.make_slot(torch.from_numpy(raws[a]), onsets[a], sl)
        assert torch.equal(aux[a - 1].cpu(), want)
    st = sharded.AntennaStream(dev, caps[0], nperseg=1024, chunk_samples=300000, slice_samples=sl, overlap=overlap,
                               aux_slots=aux)
    assert st.n_ant == 3 and st.pairs == [(0, 1), (0, 2), (1, 2)]
    outs = [st.step() for _ in range(3)]
    results, td = outs[-1].unpack()
    z = [orc.tdoa_unpack(r) for r in raws]
    want = [orc.xcorr_lag(z[j][onsets[j]:onsets[j] + sl], z[i][onsets[i]:onsets[i] + sl])[0] for i, j in td.pairs]
    assert td.lags == [int(w) for w in want]
    for (i, j), lag in zip(td.pairs, td.lags):
        assert lag + onsets[j] - onsets[i] == delays[j] - delays[i]
    assert all(m > 0.5 for m in td.margins)                # one clean peak per pair
    assert results[0].onset == onsets[0] and results[0].lag == 0
    # the kernel-packed vector (pair block included) equals the torch-packed one
    ref = sharded.pack_stream_reference(st)
    got = outs[-1][0]
    assert got.numel() == sharded.result_len(st.n_chunks, st.nperseg, 3) == ref.numel()
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6)
    # an antenna whose slice would run off the end of its capture: its pairs come back invalid
    bad = torch.tensor([n - sl + 1, 0, 0, 0], dtype=torch.int64, device="cuda")
    dev.tdoa_slot_dev(caps[2], caps[2].numel(), bad, sl, st.slots[2])
    st.tdoa()
    torch.cuda.synchronize()
    lags = st.lags.cpu().tolist()
    assert lags[0] == want[0] and lags[1] == sharded.LAG_INVALID and lags[2] == sharded.LAG_INVALID
    torch.cuda.set_stream(torch.cuda.default_stream())
    st.close()
    dev.close()
