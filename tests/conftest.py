import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes-long GPU / host-oracle tests, skipped unless OMG_RUN_SLOW=1 (tools/gpu_round_end.sh sets it): "
                                       "the driver's `pytest -m gpu` has a 20-minute limit")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("OMG_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow: OMG_RUN_SLOW=1 runs it (tools/gpu_round_end.sh)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
