import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests exercise the HIP library through its C ABI.  It normally arrives prebuilt (in-tree, __graft_entry__.build());
    on a checkout without it, compile it now -- hipcc cross-compiles gfx950 without a GPU.  There is no other code path."""
    from lcpc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
