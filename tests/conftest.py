import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """Observed parity error per GPU test (tests/parity.py): the bar is rtol 1e-3 / atol 1e-4."""
    try:
        from parity import OBSERVED
    except Exception:
        return
    if OBSERVED:
        terminalreporter.write_line("observed max |got - want| / max(1, scale) per test (bar: atol 1e-4):")
        for k, v in sorted(OBSERVED.items()):
            terminalreporter.write_line(f"  {v:.3e}  {k}")
