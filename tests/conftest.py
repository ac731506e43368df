import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """Evidence lines of the GPU suite (which chain kernels really ran): printed even with -q, so the driver's record shows them."""
    try:
        import util
    except ImportError:
        return
    for line in util.SESSION_NOTES:
        terminalreporter.write_line("jm_amd_dec: " + line)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the product library, the generator and the oracle once per session (CPU only: hipcc cross-compiles)."""
    import __graft_entry__ as ge
    ge.build()
    yield


@pytest.fixture(scope="session")
def oracle():
    from tools import streams
    return streams.Oracle()
