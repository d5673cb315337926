"""BASELINE configs[4] rehearsed on ONE GPU: two fresh processes share cuda:0, talk over gloo and
each runs the REAL per-capture pipeline (gpsjam.sharded.AntennaStream.step(): fused scan, K2, TDOA
slot, slot gather, all-pairs K5 on rank 0, result gather) on its own capture.  Rank 0's gathered
power maps, noise floors, onsets and pair lags are checked against the oracle and against the
delay the captures were built with.  Only the RCCL transport itself is not covered here (two RCCL
ranks cannot share a device); the exchange code is the one bench.py runs with backend nccl."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

N, SL, NPERSEG, CHUNK = 1_300_000, 1 << 16, 1024, 400_000
DELAYS = (0, 5)
JAM = (520_000, 1_000_000)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spec(rank):
    from gpsjam.synth import StreamSpec
    return StreamSpec(seed=17, antenna=rank, delay=DELAYS[rank], jam_start=JAM[0], jam_end=JAM[1],
                      jam_sigma=(60.0, 45.0)[rank])


def _worker(rank, world, port, overlap, q):
    for p in (os.path.join(REPO, "gps-jamming_amd"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gpsjam
        from gpsjam import sharded
        from gpsjam.synth import generate

        torch.cuda.set_device(0)
        dev = gpsjam.Device(0)
        work = torch.cuda.Stream()
        torch.cuda.set_stream(work)
        dev.set_stream(work.cuda_stream)
        raw = generate(_spec(rank), N)
        cap = torch.from_numpy(raw).cuda()
        st = sharded.AntennaStream(dev, cap, nperseg=NPERSEG, chunk_samples=CHUNK, slice_samples=SL, rank=rank,
                                   world_size=world, overlap=overlap)
        assert st.overlap is overlap
        outs = []
        for _ in range(4):                                  # several steps back to back: buffers alternate
            got = st.step()
            if rank == 0:
                res, td = got.unpack()                      # waits on the exchange's event, then D2H
                outs.append((res, td))
            else:
                assert got is None
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            res, td = outs[-1]
            for r_prev, td_prev in outs[:-1]:               # identical from step to step
                assert td_prev.lags == td.lags
                for a, b in zip(r_prev, res):
                    np.testing.assert_array_equal(a.power_map, b.power_map)
                    assert (a.onset, a.baseline, a.amp_mean) == (b.onset, b.baseline, b.amp_mean)
            q.put(("root", [dict(rank=r.rank, power_map=r.power_map.tolist(), baseline=r.baseline, threshold=r.threshold,
                                 amp_first=r.amp_first, amp_count=r.amp_count, amp_mean=r.amp_mean, onset=r.onset,
                                 lag=r.lag, ranges=r.jamming_byte_ranges(), spec=r.mean_spectrum.tolist()) for r in res],
                   dict(pairs=td.pairs, lags=td.lags, margins=td.margins)))
        else:
            q.put(("other", rank))
        st.close()
        dev.close()
    except Exception as e:                                  # surface the failure in the parent
        import traceback
        q.put(("fail", rank, repr(e) + "\n" + traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_share_one_gpu(overlap):
    import torch.multiprocessing as mp
    for p in (os.path.join(REPO, "gps-jamming_amd"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gpsjam.synth import generate
    from oracle import gpsjam_oracle as orc

    ctx = mp.get_context("spawn")                           # fresh children: no GPU state is inherited
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(m[0] != "fail" for m in msgs), msgs
    _, res, td = [m for m in msgs if m[0] == "root"][0]
    assert [r["rank"] for r in res] == [0, 1]
    onsets = []
    for rank, r in enumerate(res):
        raw = generate(_spec(rank), N)
        pm = orc.chunk_power(raw)
        np.testing.assert_allclose(np.array(r["power_map"], np.float32), pm, rtol=1e-6)
        base, thr, ranges = orc.power_threshold(pm)
        assert np.float32(r["baseline"]) == np.float32(base)
        assert r["ranges"] == [(int(a), int(b)) for a, b in ranges] and len(ranges) == 1
        k, avg = orc.rssi_amp_stats(raw, 0.0)
        assert r["amp_first"] == k and r["amp_count"] == N - k
        np.testing.assert_allclose(r["amp_mean"], avg, rtol=1e-6)
        onset = orc.tdoa_onset(orc.tdoa_unpack(raw))
        assert r["onset"] == onset
        onsets.append(onset)
        lin, _, _ = orc.widmo_waterfall(raw, nperseg=NPERSEG, chunk_samples=CHUNK)
        np.testing.assert_allclose(np.array(r["spec"], np.float32), lin.mean(axis=0), rtol=1e-4)
    assert [tuple(p) for p in td["pairs"]] == [(0, 1)]
    z = [orc.tdoa_unpack(generate(_spec(rank), N)) for rank in range(2)]
    want = orc.xcorr_lag(z[1][onsets[1]:onsets[1] + SL], z[0][onsets[0]:onsets[0] + SL])[0]
    assert td["lags"] == [int(want)]
    assert td["lags"][0] + onsets[1] - onsets[0] == DELAYS[1] - DELAYS[0]
    assert res[1]["lag"] == td["lags"][0] and res[0]["lag"] == 0
    assert td["margins"][0] > 0.5
