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
