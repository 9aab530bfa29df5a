import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, "oracle")
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The in-tree libraries are git-ignored build products: build them (hipcc cross-compiles gfx950 without a GPU; gcc for the
    oracle) when a checkout runs the tests before `__graft_entry__.build()`.  The product itself never builds on demand: it
    raises when libdsnerf_hip.so is missing."""
    need = [os.path.join(ROOT, "dual-space-nerf_amd", "libdsnerf_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    import importlib
    entry = importlib.import_module("__graft_entry__")
    entry.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
