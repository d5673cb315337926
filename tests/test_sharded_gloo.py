"""The N > 1 exchange (reference-slice broadcast + result gather + rank-0 unpacking) with two
gloo ranks on the CPU.  The per-capture numbers that the GPU kernels would produce are
supplied by the oracle here (this is a test of the distributed plumbing, which is
backend-agnostic torch.distributed code shared with bench.py's RCCL run)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (os.path.join(REPO, "gps-jamming_amd"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpsjam import sharded
        from gpsjam.synth import StreamSpec, generate
        from oracle import gpsjam_oracle as orc

        delays = (0, 5)
        n, sl, nperseg = 320000, 32768, 256
        raw = generate(StreamSpec(seed=11, antenna=rank, delay=delays[rank], jam_start=250000,
                                  jam_end=1 << 40, jam_sigma=60.0), n)
        pm = orc.chunk_power(raw)
        base, thr, ranges = orc.power_threshold(pm)
        k, avg = orc.rssi_amp_stats(raw, 0.0)
        z = orc.tdoa_unpack(raw)
        onset = orc.tdoa_onset(z)
        lin, _, _ = orc.widmo_waterfall(raw, nperseg=nperseg, chunk_samples=100000)
        cap16 = torch.from_numpy(raw.copy()).view(torch.int16)
        ref = torch.zeros(sl, dtype=torch.int16)
        if rank == 0:
            ref.copy_(cap16[onset:onset + sl])
        sharded.broadcast_reference_slice(ref, world, 0)
        ref_raw = ref.view(torch.uint8).numpy()
        lag, peak = orc.xcorr_lag(z[onset:onset + sl], orc.tdoa_unpack(ref_raw))
        vec = sharded.pack_results(
            pm.size, nperseg, torch.from_numpy(pm), torch.tensor([base, thr, float((pm > thr).sum())]),
            torch.tensor(k), torch.tensor(n - k), torch.tensor(float(avg)), torch.tensor(onset),
            torch.tensor(lag), torch.tensor(float(peak)), torch.tensor(1.0),
            torch.from_numpy(lin.mean(axis=0)), lin.shape[0], rank)
        assert vec.numel() == sharded.result_len(pm.size, nperseg)
        got = sharded.gather_results(vec, rank, world, 0)
        if rank == 0:
            res = [sharded.unpack_results(v) for v in got]
            assert [r.rank for r in res] == [0, 1]
            np.testing.assert_array_equal(res[0].power_map, pm)
            assert res[0].jamming_byte_ranges() == [(int(a), int(b)) for a, b in ranges]
            assert res[0].lag == 0 and res[0].onset == onset
            q.put(("ok", [r.lag for r in res], [r.onset for r in res], [r.amp_mean for r in res]))
        else:
            assert got is None
            q.put(("ok1", lag, onset))
        dist.barrier()
    except Exception as e:                      # surface the failure in the parent
        q.put(("fail", rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=150) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(m[0] != "fail" for m in msgs), msgs
    root = [m for m in msgs if m[0] == "ok"][0]
    other = [m for m in msgs if m[0] == "ok1"][0]
    # antenna 1 sees the burst 5 samples late; its own onset detection moves with it, and the
    # lag of its onset-aligned slice against the reference slice is what rank 0 receives
    assert root[1][1] == other[1]
    assert root[2][1] == other[2]
    assert root[2][1] - root[2][0] + root[1][1] == 5
