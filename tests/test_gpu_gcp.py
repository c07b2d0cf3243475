"""GPU: filter_gcp_distance -- the distance mask is formed on the device (alp_distance_mask) --
against the vectors captured from the reference's own function (tests/golden/gen_golden_gcp.py: g10) and the
behaviours the reference's tests/test_gcp.py::TestFilterGcpDistance pins."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_gcp.npz"))


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


def _three(xs=(100, 200, 300)):
    n = len(xs)
    return pd.DataFrame({"u": list(xs), "v": list(xs), "x": list(xs), "y": [0] * n, "z": [0] * n})


ORIGIN = {"x": 0, "y": 0, "z": 0}


def test_filter_matches_reference(L):
    from alproj_amd.gcp import filter_gcp_distance
    g = pd.DataFrame(G["filt_input"], columns=["u", "v", "x", "y", "z"], index=G["filt_input_index"])
    cam = dict(zip("xyz", G["cam"]))
    for k, (lo, hi) in enumerate(G["filt_cases"]):
        out = filter_gcp_distance(g, cam, None if np.isnan(lo) else lo, None if np.isnan(hi) else hi)
        np.testing.assert_array_equal(out.to_numpy(dtype=np.float64), G[f"filt{k}_values"])
        np.testing.assert_array_equal(out.index.to_numpy(), G[f"filt{k}_index"])


def test_filter_min_max_and_boundaries(L):
    from alproj_amd.gcp import filter_gcp_distance
    assert filter_gcp_distance(_three(), ORIGIN, min_distance=150)["x"].tolist() == [200, 300]
    assert filter_gcp_distance(_three(), ORIGIN, max_distance=250)["x"].tolist() == [100, 200]
    r = filter_gcp_distance(_three((100, 200, 300, 400)), ORIGIN, min_distance=150, max_distance=350)
    assert r["x"].tolist() == [200, 300] and list(r.index) == [0, 1]
    p = pd.DataFrame({"u": [1], "v": [1], "x": [3], "y": [4], "z": [0]})
    assert len(filter_gcp_distance(p, ORIGIN, min_distance=5)) == 1        # distance exactly 5 is kept
    assert len(filter_gcp_distance(p, ORIGIN, min_distance=5.1)) == 0


def test_filter_drops_nan_rows_and_nan_cameras(L):
    from alproj_amd.gcp import filter_gcp_distance
    g = _three()
    g["x"] = [100, np.nan, 300]
    assert filter_gcp_distance(g, ORIGIN, min_distance=0)["x"].tolist() == [100, 300]
    assert len(filter_gcp_distance(_three(), {"x": np.nan, "y": 0, "z": 0}, min_distance=0)) == 0     # NaN distance: no comparison holds


def test_distance_mask_equals_numpy_on_utm_scale_coordinates(L):
    """The mask must be numpy's bit for bit: 50 000 points at UTM magnitudes, thresholds placed ON computed distances."""
    rng = np.random.default_rng(3)
    cam = np.array([732731.25, 4051171.5, 2458.125])
    xyz = cam + rng.normal(0, 1500, (50_000, 3))
    xyz[::97, 1] = np.nan
    d = np.sqrt((xyz[:, 0] - cam[0]) ** 2 + (xyz[:, 1] - cam[1]) ** 2 + (xyz[:, 2] - cam[2]) ** 2)
    ok = ~np.isnan(xyz).any(1)
    lo, hi = float(np.nanquantile(d, 0.3)), float(np.nanquantile(d, 0.8))
    lo, hi = float(d[ok][np.argmin(np.abs(d[ok] - lo))]), float(d[ok][np.argmin(np.abs(d[ok] - hi))])    # exact distances of two rows
    with np.errstate(invalid="ignore"):
        for a, b in ((lo, None), (None, hi), (lo, hi), (None, None), (0.0, 0.0)):
            want = ok.copy()
            if a is not None:
                want &= d >= a
            if b is not None:
                want &= d <= b
            np.testing.assert_array_equal(L.distance_mask(xyz, cam, a, b), want)
    assert L.distance_mask(np.empty((0, 3)), cam, 1.0, 2.0).shape == (0,)
    with pytest.raises(L.AlprojHipError, match="min_distance must be non-negative"):
        L.distance_mask(xyz, cam, -1.0, None)
    with pytest.raises(L.AlprojHipError, match="max_distance must be >= min_distance"):
        L.distance_mask(xyz, cam, 5.0, 1.0)
