"""gpsjam.local.LocalAntennas -- every antenna of a deployment on ONE GPU, the reference's own shape (three files, one
process: GpsJammerApp/app/worker.py:97-101,586-600).  Its results must be byte for byte those of the established
single-capture pipeline (gpsjam.sharded.AntennaStream per capture + one K5 launch over their slots), and agree with the
golden vectors of the reference."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name", ["g3", "g4", "syn3", "g1"])
def test_local_antennas_are_bit_identical_to_single_streams(dev, name, graph):
    import torch
    import test_split_gpu as sg
    from gpsjam import local
    caps, kw = sg.scenarios()[name]
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    dev.set_stream(work.cuda_stream)
    try:
        want = sg._single_gpu(dev, caps, kw)
        tensors = [sg._device_range(dev, c, 0, sg._nbytes(c)) for c in caps]
        with local.LocalAntennas(dev, tensors, graph=graph, **kw) as st:
            assert len(st._sides) == min(len(caps), 3)
            for step in range(6):                              # the two result sets alternate; eager, capture, replay
                got = st.step()
                res, td = got.unpack()
                sg._assert_identical(f"{name} step {step}", sg._collect(res, td, [st.psd[a][:st.rows[a]].cpu().numpy().copy()
                                                                                  for a in range(st.n_ant)]), want)
            if graph:                                          # the steps behind the first really were graph replays
                assert st._graphs is not None and all(g is not None for g in st._graphs), "the capture was refused"
        if name == "g4":                                       # against the reference's own numbers (tests/golden)
            meta = json.load(open(os.path.join(HERE, "golden", "golden_meta.json")))
            assert [r.onset for r in res] == meta["g4"]["onset"]
            own = meta["g4"]["lags_own_start"]
            assert td.lags == [own["524288_01"], own["524288_02"], own["524288_12"]]
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        dev.set_stream(None, external=False)


def test_local_antennas_from_files(dev, tmp_path):
    """The same from capture FILES (uploaded once, resident): three antennas at the reference's slice size."""
    import torch
    import test_split_gpu as sg
    from gpsjam import local
    from gpsjam.synth import StreamSpec, generate
    n = 900_000
    raws = [generate(StreamSpec(seed=61, antenna=a, delay=d, jam_start=400_000, jam_end=800_000, jam_sigma=s), n)
            for a, (d, s) in enumerate(((0, 60.0), (7, 52.0), (-4, 56.0)))]
    paths = []
    for a, r in enumerate(raws):
        p = tmp_path / f"ant{a}.bin"
        r.tofile(p)
        paths.append(str(p))
    kw = dict(chunk_samples=131072, nperseg=1024, slice_samples=50000, rssi_threshold=0.0)
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    dev.set_stream(work.cuda_stream)
    try:
        want = sg._single_gpu(dev, raws, kw)
        st = local.from_files(dev, paths, **kw)
        res, td = st.step().unpack()
        sg._assert_identical("files", sg._collect(res, td, [st.psd[a][:st.rows[a]].cpu().numpy().copy() for a in range(3)]), want)
        # the known delays come back: lag(i, j) + onset_j - onset_i == delay_j - delay_i
        delays = (0, 7, -4)
        for (i, j), lag in zip(td.pairs, td.lags):
            assert lag + res[j].onset - res[i].onset == delays[j] - delays[i]
        st.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        dev.set_stream(None, external=False)


def test_local_antennas_edge_shapes(dev):
    """One antenna (no pairs), five antennas (more captures than side streams: chains share streams), and a capture too
    short for K4's noise span next to ordinary ones (its onset is -1, its slot invalid, its pairs GJ_LAG_INVALID as in
    the other arrangements; skrypty/triangulateTDOA.py:67-77 aborts there)."""
    import torch
    from gpsjam import local, sharded
    from gpsjam.synth import StreamSpec, generate
    n = 600_000
    mk = lambda a, d, m=n: generate(StreamSpec(seed=71, antenna=a, delay=d, jam_start=300_000, jam_end=1 << 40, jam_sigma=55.0), m)   # noqa: E731
    kw = dict(chunk_samples=131072, nperseg=1024, slice_samples=50000)
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    dev.set_stream(work.cuda_stream)
    try:
        with local.LocalAntennas(dev, [torch.from_numpy(mk(0, 0)).cuda()], **kw) as one:
            res, td = one.step().unpack()
            res2, _ = one.step().unpack()
            assert len(res) == 1 and td.pairs == [] and res[0].onset > 0 and res2[0].onset == res[0].onset
        delays = (0, 3, -2, 5, 1)
        with local.LocalAntennas(dev, [torch.from_numpy(mk(a, d)).cuda() for a, d in enumerate(delays)], **kw) as five:
            assert len(five._sides) == 3 and len(five.pairs) == 10
            for _ in range(3):
                res, td = five.step().unpack()
            for (i, j), lag in zip(td.pairs, td.lags):
                assert lag + res[j].onset - res[i].onset == delays[j] - delays[i], (i, j, lag)
        short = mk(1, 0, 150_000)                               # shorter than noise_samples + window: no onset
        with local.LocalAntennas(dev, [torch.from_numpy(mk(0, 0)).cuda(), torch.from_numpy(short).cuda(),
                                       torch.from_numpy(mk(2, 4)).cuda()], **kw) as mixed:
            for _ in range(3):
                res, td = mixed.step().unpack()
            assert res[1].onset == -1 and res[0].onset > 0 and res[2].onset > 0
            assert td.lag(0, 1) == sharded.LAG_INVALID and td.lag(1, 2) == sharded.LAG_INVALID
            assert td.lag(0, 2) + res[2].onset - res[0].onset == 4
            assert res[1].power_map.size == 5 and np.isfinite(res[1].baseline)
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
        dev.set_stream(None, external=False)
