"""`python bench.py --gpus N` starts its own ranks (VERDICT r02 missing 1): the parent never touches the GPU,
spawns torch.distributed.run as a child, passes rank 0's JSON line through and returns the ranks' status."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return env


def _json_line(stdout):
    """stdout carries ONE line, the JSON object, and nothing else (library banners -- RCCL prints one when its first
    communicator is made -- are sent to stderr by bench.py's guard_stdout)."""
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_self_launch_rendezvous_cpu():
    """No launcher on the command line: two gloo ranks find each other and rank 0's line comes through."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rendezvous-only"],
                       capture_output=True, text=True, env=_env(), timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    seats = line.pop("seats")
    assert line == {"rendezvous": "ok", "world": 2, "backend": "gloo", "sum": 3.0}
    # who sat where, all-gathered over the group itself: two ranks, two processes
    assert [d["rank"] for d in seats] == [0, 1] and len({d["pid"] for d in seats}) == 2
    assert "torch.distributed.run" in r.stderr          # the parent says what it started
    assert "diagnosis" not in r.stderr                  # nothing failed: no second run


@pytest.mark.timeout(300)
def test_failed_run_diagnoses_itself_cpu():
    """A rank dies after the group has formed (--fail-rank): the parent exits non-zero with the ranks' status and has
    run ONE fresh --rendezvous-only child whose verdict -- the group forms, so the failure is in the run -- is on
    stderr.  stdout stays free of any second JSON line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rendezvous-only", "--fail-rank", "1"],
                       capture_output=True, text=True, env=_env(), timeout=280)
    assert r.returncode != 0
    assert "--fail-rank, leaving with status 3" in r.stderr
    assert "diagnosis: one fresh --rendezvous-only run of 2 ranks over gloo" in r.stderr
    assert "diagnosis: rendezvous ok" in r.stderr and '"rendezvous": "ok"' in r.stderr
    assert "the failure above is in the run itself" in r.stderr
    assert r.stderr.count("[bench] launching") == 2     # the run and exactly one diagnosis
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) <= 1   # the failed run's own line at most


@pytest.mark.timeout(300)
def test_self_launch_relays_failure_cpu():
    """A rank that dies makes the parent exit non-zero (here: a backend that does not exist)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "no_such_backend", "--rendezvous-only"],
                       capture_output=True, text=True, env=_env(), timeout=280)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "diagnosis: rendezvous FAILED too" in r.stderr   # the same backend cannot form a group either


@pytest.mark.timeout(400)
def test_ranks_probe_rccl_in_children_and_fall_back_to_gloo_cpu():
    """`--gpus 2 --groups-only` (nccl, the default backend) on a machine where RCCL cannot form a group -- here: no GPU
    at all, and once more with the failure injected (--inject-probe-failure): every rank asks a CHILD process to form
    the RCCL group and move 1 MiB through it, the ranks agree over the gloo control group that it failed, and the
    exchange group is gloo, labelled as a fallback with the reason -- the run yields a line instead of dying with RCCL
    (VERDICT r05 "next" 2).  The line carries per_rank figures gathered over the control group."""
    for extra in ([], ["--inject-probe-failure"]):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--groups-only"] + extra,
                           capture_output=True, text=True, env=_env(), timeout=380)
        assert r.returncode == 0, r.stderr[-3000:]
        line = _json_line(r.stdout)
        assert line["groups"] == "ok" and line["world"] == 2
        assert line["transport"] == "gloo (fallback)" and line["rccl_ranks"] == 0 and line["fallback"] is True
        assert line["exchange_backend"] == "gloo" and line["control_backend"] == "gloo"
        assert line["rccl_probe"]["ok"] is False and all(st != 0 for st in line["rccl_probe"]["statuses"])
        assert "RCCL probe failed on rank(s) [0, 1]" in line["fallback_of"]
        if extra:
            assert "status 3" in line["fallback_of"] and "--probe-fail" in line["fallback_of"]
        pr = line["per_rank"]
        assert pr["k2_ms"] == {"min": 1.0, "max": 2.0, "by_rank": [1.0, 2.0]} and pr["exchange_ms"]["by_rank"] == [0.5, 1.0]
        assert r.stderr.count("[bench] launching") == 1      # the ranks rescued themselves: no second run


@pytest.mark.timeout(400)
def test_launcher_runs_once_more_over_gloo_when_the_nccl_run_dies_cpu():
    """Second line of defence: the ranks trusted RCCL (--no-probe) and died with it.  The launcher holds the failed
    run's output back, diagnoses (the nccl group cannot form), starts ONE more fresh child over gloo with the reason in
    --fallback-of, passes ITS line on and returns 0 because that run succeeded."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--groups-only", "--no-probe"],
                       capture_output=True, text=True, env=_env(), timeout=380)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["transport"] == "gloo (fallback)" and line["rccl_ranks"] == 0 and line["fallback"] is True
    assert "the run over nccl ended with status" in line["fallback_of"] and "the group cannot form" in line["fallback_of"]
    assert "per_rank" in line and "rccl_probe" not in line
    assert r.stderr.count("[bench] launching") == 3          # the run, its diagnosis, the fallback run
    assert "fallback: one fresh run of 2 ranks over gloo" in r.stderr


def test_parent_does_not_import_torch_before_launch():
    """The launching parent must not initialise the GPU: the launch happens before torch / gpsjam are imported."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("import torch")
    assert "os.exec" not in src and "execv" not in src


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_self_launch_two_ranks_share_gpu():
    """`python bench.py --gpus 2 --backend gloo --share-gpu --steps 3`: the real step on two ranks that share
    cuda:0, started by bench.py itself; one line, n_gpus 2, self_check passed."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "3",
                        "--warmup", "1", "--precondition", "2", "--cpu-sample-chunks", "1"],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3
    assert line["self_check"]["passed"] is True, line["self_check"]
    assert line["self_check"]["pairs_checked"] == 1 and line["self_check"]["streams_checked"] == 2
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # what the line says about the exchange is read from the live group: gloo carried it (no RCCL rank), two ranks were
    # seen, they sat on ONE physical device -- a rehearsal, and the line says so
    assert line["rccl_ranks"] == 0 and line["ranks_seen"] == 2 and line["rehearsal"] is True
    assert [d["rank"] for d in line["devices"]] == [0, 1] and len({d["pid"] for d in line["devices"]}) == 2
    assert line["distinct_devices"] == 1 and line["devices"][0]["pci"] == line["devices"][1]["pci"]
    assert line["self_check"]["devices_ok"] is True     # --share-gpu announced it
    # every rank's own figures, gathered after the timed region (VERDICT r05 "next" 2)
    pr = line["per_rank"]
    for k in ("k2_ms", "scan_ms", "exchange_ms", "step_ms"):
        assert len(pr[k]["by_rank"]) == 2 and 0 < pr[k]["min"] <= pr[k]["max"], (k, pr[k])
    assert line["exchange_backend"] == "gloo" and "fallback" not in line


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_fall_back_to_gloo_when_the_rccl_probe_fails():
    """`--gpus 2 --share-gpu --inject-probe-failure --steps 3` over nccl (the default): the probe children report
    failure, the ranks agree on gloo and the REAL step runs -- one line, transport "gloo (fallback)", rccl_ranks 0,
    fallback_of, per_rank, self_check passed, status 0."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--inject-probe-failure", "--steps", "3",
                        "--warmup", "1", "--precondition", "2", "--cpu-sample-chunks", "1", "--capture-bytes", str(1 << 28)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["transport"] == "gloo (fallback)" and line["config"]["backend"] == "gloo"
    assert line["rccl_ranks"] == 0 and line["fallback"] is True and "RCCL probe failed" in line["fallback_of"]
    assert line["rehearsal"] is True and line["self_check"]["passed"] is True, line["self_check"]
    assert len(line["per_rank"]["k2_ms"]["by_rank"]) == 2 and line["per_rank"]["exchange_ms"]["max"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rccl_ranks_on_one_gpu_do_not_pass_as_a_two_gpu_run():
    """`--gpus 2 --share-gpu` WITHOUT `--backend gloo`: two RCCL ranks cannot share a device.  Whatever the library makes
    of it -- refusing the communicator, or timing out -- the run must end non-zero and must not print a line that reads
    as a two-GPU result."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--rendezvous-only", "--no-diagnosis",
                        "--launch-timeout", "240"], capture_output=True, text=True, env=_env(), timeout=560)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode != 0, (r.stdout[-500:], r.stderr[-1500:])
    assert not any('"rendezvous": "ok"' in ln or '"n_gpus": 2' in ln for ln in lines), lines


def test_self_check_refuses_ranks_on_one_device():
    """N ranks that report fewer than N physical devices without --share-gpu fail the line's self_check."""
    sys.path.insert(0, REPO)
    import bench
    two_on_one = {"devices": [{"rank": 0, "pci": "0000:05:00.0", "uuid": "a"}, {"rank": 1, "pci": "0000:05:00.0", "uuid": "a"}],
                  "distinct_devices": 1}
    assert bench.self_check([], type("T", (), {"pairs": [], "lags": []})(), [], 0, 2, 0, proof=two_on_one, world=2)["devices_ok"] is False
    assert bench.self_check([], type("T", (), {"pairs": [], "lags": []})(), [], 0, 2, 0, proof=two_on_one, world=2,
                            share_gpu=True)["devices_ok"] is True
    two_on_two = {"devices": [{"rank": 0, "pci": "0000:05:00.0"}, {"rank": 1, "pci": "0000:06:00.0"}], "distinct_devices": 2}
    assert bench.self_check([], type("T", (), {"pairs": [], "lags": []})(), [], 0, 2, 0, proof=two_on_two, world=2)["devices_ok"] is True
    lost_rank = {"devices": [{"rank": 0, "pci": "a"}, {"rank": 0, "pci": "b"}], "distinct_devices": 2}
    assert bench.self_check([], type("T", (), {"pairs": [], "lags": []})(), [], 0, 2, 0, proof=lost_rank, world=2)["devices_ok"] is False
    ident = bench.parse_identity("rank=3 pid=77 host=box pci=0000:05:00.0 uuid=00ff hip=2 torch=x/0000:05:00")
    assert ident == {"rank": 3, "pid": 77, "host": "box", "pci": "0000:05:00.0", "uuid": "00ff", "hip": 2, "torch": "x/0000:05:00"}


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_split_bench_two_ranks_share_gpu():
    """`python bench.py --gpus 2 --split`: three captures cut over two ranks (sharing cuda:0, gloo), strong scaling
    line, self_check passed (known delays and burst spans come back through the split + combine)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--split", "--backend", "gloo", "--share-gpu", "--steps", "3",
                        "--warmup", "1", "--precondition", "2", "--capture-bytes", str(1 << 28)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["antennas"] == 3
    assert line["self_check"]["passed"] is True, line["self_check"]
    assert line["self_check"]["pairs_checked"] == 3 and line["self_check"]["streams_checked"] == 3
    assert len({p[5] for p in line["config"]["parts"]}) == 2


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_collective_path_over_rccl_on_one_gpu():
    """`python bench.py --gpus 1 --force-exchange`: a process group of ONE over the nccl backend (= RCCL) -- two RCCL
    ranks cannot share a device, so this is as far as one GPU goes: the communicator is bootstrapped over the loopback
    interface, and the slot all-gather (all_gather_into_tensor), the result gather (gather into row views), the
    barriers and the MAX all-reduce of the N > 1 path are issued on the pipeline's streams exactly as there.  The
    line must carry the same results as the plain N = 1 run (self_check: known delays, burst spans)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-exchange", "--steps", "3", "--warmup", "1",
                        "--precondition", "2", "--no-cpu-baseline", "--no-end-to-end", "--capture-bytes", str(1 << 28)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["forced_exchange"] is True
    # read from the live group (dist.get_world_size() of an nccl group), with the one device it ran on
    assert line["rccl_ranks"] == 1 and line["ranks_seen"] == 1 and line["distinct_devices"] == 1
    assert "over nccl" in line["identity_exchanged_via"] and line["rehearsal"] is False
    d = line["devices"][0]
    assert d["rank"] == 0 and d["pci"].count(":") == 2 and d["pid"] > 0 and d["hip"] == 0
    assert line["self_check"]["passed"] is True, line["self_check"]
    assert line["self_check"]["pairs_checked"] == 3 and line["results"]["lags"] and line["value"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_groups_of_an_n_gpu_run_on_one_gpu():
    """`--gpus 1 --force-exchange --mixed-groups`: the groups exactly as N > 1 ranks form them -- gloo for control (barriers,
    the MAX all-reduce of the step time, per_rank), an RCCL group of its own (`dist.new_group(backend="nccl", device_id=...)`)
    handed to the pipeline for the slot all-gather and the result gather -- with the one rank a single GPU allows.  What a
    first run on a real node would otherwise meet for the first time: the mixed-backend calls themselves."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-exchange", "--mixed-groups", "--steps", "3", "--warmup", "1",
                        "--precondition", "2", "--no-cpu-baseline", "--no-end-to-end", "--no-reference-point",
                        "--capture-bytes", str(1 << 28)], capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["control_backend"] == "gloo" and line["exchange_backend"] == "nccl"
    assert line["rccl_ranks"] == 1 and "over nccl" in line["identity_exchanged_via"] and line["rehearsal"] is False
    assert line["self_check"]["passed"] is True, line["self_check"]
    assert len(line["per_rank"]["k2_ms"]["by_rank"]) == 1 and line["per_rank"]["exchange_ms"]["max"] > 0
    # and the split pipeline through the same groups
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--split", "--force-exchange", "--mixed-groups", "--steps", "2",
                        "--warmup", "1", "--precondition", "2", "--capture-bytes", str(1 << 27)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["control_backend"] == "gloo" and line["exchange_backend"] == "nccl" and line["rccl_ranks"] == 1
    assert line["self_check"]["passed"] is True, line["self_check"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_per_rank_load_of_eight_antennas_on_one_gpu():
    """`--emulate-world 8 --force-exchange`: rank 0's share of an eight-antenna deployment (seven further slots, 4 of the
    28 pairs in one K5 launch, both collectives over the one-rank RCCL group) -- the lags must come out at the delays
    the captures were built with."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--emulate-world", "8", "--force-exchange", "--steps", "2",
                        "--warmup", "1", "--precondition", "2", "--no-cpu-baseline", "--no-end-to-end",
                        "--capture-bytes", str(1 << 28)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["emulated_world"] == 8 and line["config"]["xcorr_antennas"] == 8 and line["config"]["xcorr_pairs"] == 4
    assert line["results"]["pairs"] == [[0, 1], [0, 2], [0, 3], [0, 4]]
    assert line["self_check"]["passed"] is True and line["self_check"]["pairs_checked"] == 4, line["self_check"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_split_collective_path_over_rccl_on_one_gpu():
    """`--split --force-exchange` at N = 1: the slot all-gather and the part-vector gather of the split path over the
    one-rank RCCL group; results as in the plain split run (self_check)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--split", "--force-exchange", "--steps", "2", "--warmup", "1",
                        "--precondition", "2", "--capture-bytes", str(1 << 27)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["scaling"] == "strong" and line["forced_exchange"] is True and line["rccl_ranks"] == 1
    assert line["self_check"]["passed"] is True, line["self_check"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_split_rank0_load_of_eight_gpus_on_one():
    """`--split --emulate-world 8 --force-exchange`: rank 0 of eight on ONE GPU -- its own eighth of the three captures, its
    share of the pairs, both collectives over the one-rank RCCL group, and the three-launch combine over ALL eight ranks'
    part vectors (the other seven's prepared before the timed region).  Known delays and burst spans come back."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--split", "--emulate-world", "8", "--force-exchange", "--steps", "3",
                        "--warmup", "1", "--precondition", "2", "--capture-bytes", str(1 << 28)],
                       capture_output=True, text=True, env=_env(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["scaling"] == "strong" and line["emulated_world"] == 8 and line["projected"] is True and line["rehearsal"] is True
    assert len({p[5] for p in line["config"]["parts"]}) == 8 and len(line["config"]["parts"]) == 10
    assert line["config"]["own_bytes_rank0"] * 8 <= 3 * (1 << 28) + 8 * 8192000
    assert line["self_check"]["passed"] is True and line["self_check"]["pairs_checked"] == 3, line["self_check"]
    ch = line["second_stream_chain"]
    assert ch["combine_launches"] == 3 and ch["k2_ms"] > 0 and ch["pack_gather_combine_ms"] > 0
