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


@pytest.fixture(scope="session", autouse=True)
def abi_call_coverage():
    """ALP_ABI_COVERAGE=<file>: record which entry points of the C ABI the session calls (every ctypes function of
    alproj_amd._lib is wrapped by a counting closure) and write the ones that were never called.  Off by default."""
    out = os.environ.get("ALP_ABI_COVERAGE")
    if not out:
        yield
        return
    from alproj_amd import _lib
    lib = _lib.load()
    calls = {name: 0 for name in _lib._SIGNATURES}

    def wrap(name, fn):
        def counted(*a):
            calls[name] += 1
            return fn(*a)
        return counted

    for name in calls:
        setattr(lib, name, wrap(name, getattr(lib, name)))
    yield
    never = sorted(n for n, c in calls.items() if c == 0)
    with open(out, "w") as f:
        f.write(f"{len(calls) - len(never)} of {len(calls)} entry points of the C ABI were called by this session; never called: {never}\n")
        for n in sorted(calls):
            f.write(f"{calls[n]:9d}  {n}\n")


@pytest.fixture(scope="session", autouse=True)
def python_line_coverage():
    """ALP_PY_COVERAGE=<file>: which lines of alproj_amd/*.py the session executes (a sys.settrace collector restricted to the
    package's files; no coverage tool is installed here).  Writes "path:line" of every executed line; tools/py_coverage.py
    merges several such files and lists the lines never executed.  Off by default."""
    out = os.environ.get("ALP_PY_COVERAGE")
    if not out:
        yield
        return
    import threading
    pkg = os.path.join(ROOT, "alproj_amd") + os.sep
    seen = set()

    def local(frame, event, arg):
        if event == "line":
            seen.add((frame.f_code.co_filename, frame.f_lineno))
        return local

    def tracer(frame, event, arg):
        if frame.f_code.co_filename.startswith(pkg):
            seen.add((frame.f_code.co_filename, frame.f_lineno))
            return local
        return None

    sys.settrace(tracer)
    threading.settrace(tracer)
    yield
    sys.settrace(None)
    threading.settrace(None)
    with open(out, "w") as f:
        for path, line in sorted(seen):
            f.write(f"{os.path.relpath(path, ROOT)}:{line}\n")
