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


# --- wall-time budget of the GPU suite -------------------------------------------------------------------------------
# The driver gives `pytest -m gpu` 1 200 s; past that every row counts as untested.  The suite is to stay well inside
# (target <= 450 s on the driver's box; ~170 s on a dev box): a session that runs GPU tests for longer than
# MISO_GPU_SUITE_BUDGET_S (default 900) FAILS, loudly, so that the builder sees it a round before the driver does.
import time as _time

_T0 = _time.time()


def pytest_sessionfinish(session, exitstatus):
    import torch
    budget = float(os.environ.get("MISO_GPU_SUITE_BUDGET_S", "900"))
    took = _time.time() - _T0
    ran_gpu = torch.cuda.is_available() and any("gpu" in it.keywords for it in getattr(session, "items", []))
    if ran_gpu and took > budget:
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        msg = (f"GPU suite took {took:.0f} s > budget {budget:.0f} s (driver limit 1 200 s): trim it "
               f"(pytest --durations=40) before adding tests")
        if tr is not None:
            tr.write_line("ERROR: " + msg, red=True)
        else:
            print("ERROR: " + msg, file=sys.stderr)
        session.exitstatus = 1
