import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU run.  The driver runs `pytest -x`, so the first red test hides everything behind it:
# the oracle comparisons come FIRST (device KATs and the reference's recorded runs, the seeded fuzz against the C
# oracle, config 5 against the reference's trained nets, the full-size configs), then the kernels against autograd,
# then the wider entry points, and the multi-process / timing-sensitive files LAST.  Files not named keep their
# alphabetical order between the two groups.
_FIRST = ["test_gpu_parity.py", "test_gpu_quirks.py", "test_gpu_fuzz.py", "test_gpu_nn.py", "test_gpu_nn_full.py", "test_gpu_large.py",
          "test_gpu_api.py", "test_gpu_contnet.py", "test_gpu_calibrator.py", "test_gpu_localvol.py", "test_gpu_mlp.py"]
_LAST = ["test_gpu_step_multi.py", "test_gpu_dist.py", "test_gpu_multirank.py", "test_gpu_facade_ranks.py",
         "test_gpu_nn_dist.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)

    items.sort(key=rank)  # stable: the order inside a file, and of unnamed files, is unchanged


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
