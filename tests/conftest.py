import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["oracle-cpu", pytest.param("hip", marks=pytest.mark.gpu)])
def device_backend(request, monkeypatch):
    """Device string for tests of the host-side mirror: on CPU the HIP operators are replaced
    by the oracle (test-only stand-ins, tests/oracle_backend.py); on the GPU box the real
    library runs."""
    if request.param == "oracle-cpu":
        import oracle_backend
        oracle_backend.install(monkeypatch)
        return "cpu"
    return "cuda:0"
