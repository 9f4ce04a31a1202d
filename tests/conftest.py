import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a machine without a GPU is skipped rather than failed,
    # so that a plain `pytest tests/` works in the CPU container.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import bd_oracle
    bd_oracle.lib()
    return bd_oracle


@pytest.fixture(scope="session")
def bd():
    import beat_detector_amd
    beat_detector_amd.lib()
    return beat_detector_amd


@pytest.fixture
def knobs(monkeypatch, bd):
    """Experiment switches of the library (BD_* environment variables).  The library reads its
    environment once, so every change is published with bd_tuning_reload, the restored environment
    at the end of the test included."""
    L = bd.lib()

    class Knobs:
        def set(self, name, value):
            monkeypatch.setenv(name, str(value))
            L.bd_tuning_reload()

        def unset(self, name):
            monkeypatch.delenv(name, raising=False)
            L.bd_tuning_reload()

    yield Knobs()
    monkeypatch.undo()
    L.bd_tuning_reload()


@pytest.fixture(scope="session")
def oracle_ext():
    from oracle import bd_oracle_ext
    bd_oracle_ext.lib()
    return bd_oracle_ext


def pytest_terminal_summary(terminalreporter):
    """Band values are held to 1e-4 relative, with no absolute floor: say how many were compared."""
    from tests import util
    st = util.FLOOR_STATS
    if st["values"]:
        terminalreporter.write_line(
            f"band values compared: {st['values']}, each within 1e-4 relative of the oracle's (no absolute floor)")
