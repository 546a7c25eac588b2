import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    return {
        "paths": np.load(os.path.join(GOLDEN, "paths_features.npz")),
        "poly": np.load(os.path.join(GOLDEN, "poly_flows.npz")),
        "nn": np.load(os.path.join(GOLDEN, "v3_frozen_nn.npz")),
        "per_step": np.load(os.path.join(GOLDEN, "per_step_ref.npz")),
        "nn_heston_put": np.load(os.path.join(GOLDEN, "v3_frozen_nn_heston_put.npz")),
        "scalars": json.load(open(os.path.join(GOLDEN, "scalars.json"))),
    }


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session (one process on the card)."""
    from options_model_amd import _ffi

    if _ffi.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the GPU box")
    c = _ffi.Context(0)
    yield c
    c.close()
