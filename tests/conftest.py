import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (os.path.join(REPO, "gps-jamming_amd"), os.path.join(HERE, "golden"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The order the suite runs in (VERDICT r04 "next" 2).  The driver runs `pytest -x`; files used to run alphabetically,
# which put the thread-killing C programs (test_c_abi_gpu) and the process-spawning launcher rehearsals
# (test_bench_launch) IN FRONT of the parity tests of the hot path -- one peripheral failure at position 15 left 359
# tests, the Welch kernel's among them, unreached in GPUTEST_r04.  Core first: (1) golden / oracle parity of K1-K5 and
# the BASELINE configs, (2) the "next" rows and the later rounds' additions, (3) sharded / split rehearsals in one
# process, (4) tests that spawn processes, (5) last, robustness: error paths, threads, killed callers.  This is
# ordering only: nothing is deselected, skipped or marked xfail, and the CPU tests (stage 0) keep their place in front.
SUITE_ORDER = [
    # 1 -- parity of the hot path
    "test_gpu_parity.py", "test_baseline_configs_gpu.py", "test_dropin_modules.py", "test_extremes_gpu.py",
    "test_gpu_random_sweep.py", "test_large_capture_gpu.py",
    # 2 -- next rows, later additions
    "test_acq_gpu.py", "test_local_gpu.py", "test_round2_gpu.py", "test_round3_gpu.py", "test_round4_gpu.py",
    "test_round5_gpu.py", "test_round6_gpu.py",
    # 3 -- several ranks' work in one process
    "test_sharded_gpu.py", "test_sharded_world8_gpu.py", "test_split_gpu.py", "test_split_random_gpu.py",
    # 4 -- child processes (launchers, two ranks on one GPU)
    "test_sharded_two_rank_gpu.py", "test_bench_launch.py",
    # 5 -- robustness: error paths, host threads, callers killed inside the library
    "test_errors_gpu.py", "test_threads_gpu.py", "test_c_abi_gpu.py",
]
#: within the last file, the test that kills threads runs at the very end
LAST_TESTS = ["test_abandoned_callers_do_not_block_the_context"]


def suite_rank(item) -> tuple:
    """Sort key of a collected test: (stage of its file, is-it-one-of-the-last).  Unknown GPU files go between stage 4
    and 5 (a new file must not jump in front of the parity tests by accident); the sort is stable, so tests keep their
    order within a file."""
    fname = os.path.basename(str(item.fspath))
    is_gpu = item.get_closest_marker("gpu") is not None
    if not is_gpu:
        return (-1, 0)
    try:
        stage = SUITE_ORDER.index(fname)
    except ValueError:
        stage = len(SUITE_ORDER) - 3.5
    return (stage, 1 if item.name.split("[")[0] in LAST_TESTS else 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=suite_rank)


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(HERE, "golden", "golden_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def g1_raw():
    import golden_inputs
    return golden_inputs.g1_stream()


@pytest.fixture(scope="session")
def g2_raw():
    import golden_inputs
    return golden_inputs.g2_stream()


@pytest.fixture(scope="session")
def g3_raws():
    import golden_inputs
    return golden_inputs.g3_streams()


@pytest.fixture(scope="session")
def g4_raws():
    import golden_inputs
    return golden_inputs.g4_streams()


@pytest.fixture(scope="session")
def dev():
    """One library context on cuda:0 for the GPU parity tests (fails loudly if the HIP
    library is missing -- there is no CPU fallback)."""
    import gpsjam
    d = gpsjam.Device(0)
    yield d
    d.close()


def sha256(a: np.ndarray) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
