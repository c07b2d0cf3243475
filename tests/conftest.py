import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The tests exercise the in-tree libalproj_hip.so; (re)build it when it is missing or older
    than its sources (hipcc cross-compiles without a GPU).  The product itself never builds or
    falls back on demand: alproj_amd._lib raises when the library is absent."""
    from alproj_amd import _build
    try:
        _build.build()
    except Exception as e:            # no hipcc on this machine: the tests that need the library will say so
        print(f"conftest: could not build libalproj_hip.so: {e}", file=sys.stderr)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
