import os
import sys

import pytest

# a glibc abort message (heap check, assert) goes to stderr, i.e. into the test log, not to a terminal
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Make sure the in-tree libraries exist (they travel to the GPU box prebuilt)."""
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "myldpccppapi_amd", "libldpc_hip.so")):
        ge.build()
    return True
