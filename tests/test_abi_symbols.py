"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
that include/alproj_hip.h declares, the ctypes table covers the same set, and the product
path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "alproj_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(alp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from alproj_amd import _build, _lib
    _build.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_declares_functions():
    names = header_functions()
    assert "alp_project" in names and "alp_eval_population" in names and "alp_render" in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol(built_lib):
    for name in header_functions():
        assert hasattr(built_lib, name), f"{name} declared in alproj_hip.h but not exported"


def test_ctypes_table_matches_header():
    from alproj_amd import _lib
    assert sorted(_lib._SIGNATURES) == header_functions()
    lib = _lib.load()
    version = int(re.search(r"#define ALP_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.alp_abi_version() == version == 7


def test_header_cites_reference_for_each_entry_point():
    src = open(HEADER).read()
    for token in ("src/alproj/optimize.py:122-155", "src/alproj/optimize.py:215-237",
                  "src/alproj/optimize.py:420-423", "src/alproj/project.py:145-294",
                  "src/alproj/project.py:111-143", "src/alproj/optimize.py:157-178"):
        assert token in src


def _have_gpu():
    from alproj_amd import _lib
    n = ctypes.c_int()
    _lib.load().alp_device_count(ctypes.byref(n))
    return n.value > 0


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device every compute entry point must refuse, loudly."""
    if _have_gpu():
        pytest.skip("a GPU is present")
    import numpy as np
    import pandas as pd
    from alproj_amd import _lib
    from alproj_amd import optimize as opt
    with pytest.raises(_lib.AlprojHipError) as e:
        _lib.init(0)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)
    df = pd.DataFrame(np.zeros((4, 3)), columns=["x", "y", "z"])
    p = {k: 1.0 for k in _lib.PARAM_KEYS}
    with pytest.raises(_lib.AlprojHipError):
        opt.project(df, p)
    with pytest.raises(_lib.AlprojHipError):
        opt.rmse(pd.DataFrame(np.zeros((4, 2)), columns=["u", "v"]), pd.DataFrame(np.zeros((4, 2)), columns=["u", "v"]))
    # un-initialised library: ALP_ENOTINIT
    lib = _lib.load()
    assert lib.alp_synchronize() == -2


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "alproj_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "oracle/" not in text, f


def test_shipped_library_has_no_development_switch():
    """the render translation unit keeps timing / census / stage-skipping builds, several of which draw wrong images
    by design, behind -DALP_DEV; the library the tests and the bench load must have been compiled with none"""
    from alproj_amd import _lib
    assert _lib.build_flags() == ""
    csrc = os.path.join(ROOT, "alproj_amd", "csrc")
    src = open(os.path.join(csrc, "alp_raster.hip")).read()
    head = src[:src.index("namespace alp {")]
    stages = re.findall(r'#include "(raster_[a-z]+\.h)"', src)
    assert len(stages) >= 6
    for name in ["alp_raster.hip"] + stages:
        text = open(os.path.join(csrc, name)).read()
        for switch in re.findall(r"#\s*(?:if|elif)(?:def|ndef)?\s+(?:!?defined\()?([A-Z][A-Z0-9_]+)", text):
            if switch.startswith(("ALP_DEV", "__")):
                continue
            assert switch in head, f"{switch} is used in {name} but not listed in the development-switch guard of alp_raster.hip"
