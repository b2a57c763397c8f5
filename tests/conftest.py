import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite: the parity tests against the reference's golden fixtures and the oracle (whole models, every config,
# the "next" rows, the exchange steps) run FIRST, the bench-path and kernel-level tests after them, so that under `-x` one late
# failure cannot hide the parity evidence (round 2: one noisy assertion left 302 of 325 tests unreached).
_ORDER = ["test_model_gpu", "test_v1_gpu", "test_downstream_gpu", "test_validation_gpu", "test_trainer_gpu", "test_input_u8_gpu",
          "test_dist_gpu", "test_comm_gpu", "test_bench_dist_gpu", "test_bench_path_gpu", "test_kernels_gpu"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER)
    items.sort(key=rank)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
