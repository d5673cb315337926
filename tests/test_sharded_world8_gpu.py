"""BASELINE configs[4] -- eight antennas, one capture per rank -- with the REAL kernels of every rank on ONE GPU, in
one process.  Eight `AntennaStream`s (rank r of a world of 8 each) exchange through an in-process stand-in for the
communicator (`transport=` accepts any object with `allgather` / `gather`): the slot of rank r goes through the
kernels, lands in the shared slot table, every rank solves ITS share of the 28 antenna pairs with one multi-pair K5
launch over the eight slots, packs its result vector with the library's kernel, and rank 0 unpacks the gathered rows.
What the gloo tests at world 8 cannot show (their K5 is the CPU twin) and two-process tests cannot reach (at most six
processes per GPU here): pair dealing with antenna indices up to 7, pair capacity 4, rows of eight ranks.
Reference: the pair-wise solve of skrypty/triangulateTDOA.py:60-90, generalised from two antennas to N."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (os.path.join(REPO, "gps-jamming_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

W = 8
N, SL, NPERSEG, CHUNK = 1_300_000, 1 << 16, 1024, 400_000
DELAYS = (0, 3, -5, 7, -2, 4, -6, 1)
JAM = (520_000, 1_000_000)


class LocalComm:
    """In-process stand-in for gpsjam.comm.Communicator: ranks run one after the other, the 'wire' is a tensor."""

    def __init__(self, dev, rank, wire):
        self.dev, self.rank, self.wire = dev, rank, wire

    def allgather(self, my_slot, nbytes, slots):
        assert nbytes == my_slot.numel() and slots.shape[0] == W
        self.wire["slots"][self.rank].copy_(my_slot)
        slots.copy_(self.wire["slots"])          # ranks run in two sweeps: by the second one every slot is on the wire

    def gather(self, vec, nbytes, rows, dst):
        assert dst == 0 and nbytes == vec.numel() * vec.element_size()
        self.wire["rows"][self.rank].copy_(vec)
        if rows is not None:
            rows.copy_(self.wire["rows"])

    def close(self):
        pass


@pytest.mark.timeout(600)
def test_eight_ranks_worth_of_kernels_on_one_gpu():
    import torch
    import gpsjam
    from gpsjam import sharded
    from gpsjam.synth import StreamSpec, generate
    from oracle import gpsjam_oracle as orc

    dev = gpsjam.Device(0)
    work = torch.cuda.Stream()
    torch.cuda.set_stream(work)
    dev.set_stream(work.cuda_stream)
    specs = [StreamSpec(seed=23, antenna=a, delay=DELAYS[a], jam_start=JAM[0], jam_end=JAM[1],
                        jam_sigma=60.0 * (1.0, 0.9, 0.85, 0.95)[a % 4]) for a in range(W)]
    raws = [generate(s, N) for s in specs]
    caps = [torch.from_numpy(r).cuda() for r in raws]
    sb = dev.tdoa_slot_bytes(SL)
    wire = {"slots": torch.zeros((W, sb), dtype=torch.uint8, device="cuda"), "rows": None}
    streams = []
    for r in range(W):
        st = sharded.AntennaStream(dev, caps[r], nperseg=NPERSEG, chunk_samples=CHUNK, slice_samples=SL, rank=r,
                                   world_size=W, overlap=False, transport=LocalComm(dev, r, wire))
        assert st.pairs == sharded.pairs_of_rank(r, W) and st.pair_cap == 4 and st.n_ant == W
        streams.append(st)
    wire["rows"] = torch.zeros((W, streams[0].result.numel()), dtype=torch.float64, device="cuda")
    assert sorted(sharded.canonical_pair(i, j, 0)[:2] for st in streams for i, j in st.pairs) == sharded.all_pairs(W)

    got = None
    for sweep in range(2):                        # sweep 0 puts every slot on the wire, sweep 1 is the real step
        for r in list(range(1, W)) + [0]:         # rank 0 last: it gathers
            st = streams[r]
            st.scan()
            st.tdoa()
            out = st.exchange(0)
            assert (out is None) == (r != 0)
            if r == 0:
                got = out
    torch.cuda.synchronize()
    res, td = got.unpack()
    assert [x.rank for x in res] == list(range(W)) and td.pairs == sharded.all_pairs(W) and len(td.lags) == 28

    onsets = []
    for a, x in enumerate(res):
        pm = orc.chunk_power(raws[a])
        np.testing.assert_allclose(x.power_map, pm, rtol=1e-6)
        base, _, ranges = orc.power_threshold(pm)
        assert np.float32(x.baseline) == np.float32(base) and x.jamming_byte_ranges() == [(int(p), int(q)) for p, q in ranges]
        onset = orc.tdoa_onset(orc.tdoa_unpack(raws[a]))
        assert x.onset == onset and x.amp_count == N and x.amp_first == 0
        onsets.append(onset)
    # every pair: the delay the captures were built with, and the oracle's lag on the same slices for a sample of them
    for (i, j), lag, m in zip(td.pairs, td.lags, td.margins):
        assert lag != sharded.LAG_INVALID and lag + onsets[j] - onsets[i] == DELAYS[j] - DELAYS[i], (i, j, lag)
        assert m > 0.3
    z = {a: orc.tdoa_unpack(raws[a]) for a in (0, 3, 4, 7)}
    for i, j in ((0, 4), (3, 7), (4, 7), (0, 7)):
        want = orc.xcorr_lag(z[j][onsets[j]:onsets[j] + SL], z[i][onsets[i]:onsets[i] + SL])[0]
        assert td.lag(i, j) == int(want)
    for a in range(1, W):
        assert res[a].lag == td.lag(0, a)
    for st in streams:
        st.close()
    dev.close()
