"""set_gcp / filter_gcp_distance against vectors captured from the
reference's own functions (tests/golden/gen_golden_gcp.py), plus the behaviours the
reference's tests/test_gcp.py::TestFilterGcpDistance pins.  set_gcp's DataFrame join and the
validation / nothing-to-filter returns of filter_gcp_distance run on the CPU; the device paths
(ReverseProjection -> alp_render_gather, the distance mask alp_distance_mask) are in the gpu-marked
tests/test_gpu_gcp.py and tests/test_gpu_raster.py."""
import os

import numpy as np
import pandas as pd
import pytest

from alproj_amd.gcp import filter_gcp_distance, set_gcp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_gcp.npz"))


def _rev():
    idx = G["rev_index"]
    w = int(G["w"])
    xyz = G["rev_xyz"]
    return pd.DataFrame({"u": (idx % w).astype("int16"), "v": (idx // w).astype("int16"),
                         "x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]}, index=idx)


def _match(key):
    m = G[key]
    return pd.DataFrame({"u_org": m[:, 0], "v_org": m[:, 1], "u_sim": m[:, 2], "v_sim": m[:, 3]})


@pytest.mark.parametrize("key,exp", [("match", "set"), ("match_f", "setf")])
@pytest.mark.filterwarnings("ignore:You are merging on int and float")
def test_set_gcp_frame_matches_reference(key, exp):
    out = set_gcp(_match(key), _rev())
    assert list(out.columns) == list(G["set_columns"])
    np.testing.assert_array_equal(out.index.to_numpy(), G[f"{exp}_index"])
    np.testing.assert_array_equal(out.to_numpy(dtype=np.float64), G[f"{exp}_values"])


def _three(xs=(100, 200, 300)):
    n = len(xs)
    return pd.DataFrame({"u": list(xs), "v": list(xs), "x": list(xs), "y": [0] * n, "z": [0] * n})


ORIGIN = {"x": 0, "y": 0, "z": 0}


def test_filter_empty_copy_nan():
    e = pd.DataFrame(columns=["u", "v", "x", "y", "z"])
    r = filter_gcp_distance(e, ORIGIN, min_distance=100)
    assert len(r) == 0 and list(r.columns) == ["u", "v", "x", "y", "z"]
    g = _three((100,))
    r = filter_gcp_distance(g, ORIGIN)
    r.iloc[0, 0] = 999
    assert g.iloc[0, 0] == 100                                              # a copy, not a view


def test_filter_validation():
    g = _three((100,))
    for missing in "xyz":
        prm = {k: 0 for k in "xyz" if k != missing}
        with pytest.raises(KeyError, match=f"params must contain '{missing}' key"):
            filter_gcp_distance(g, prm, min_distance=100)
    with pytest.raises(ValueError, match="min_distance must be non-negative"):
        filter_gcp_distance(g, ORIGIN, min_distance=-10)
    with pytest.raises(ValueError, match="max_distance must be >= min_distance"):
        filter_gcp_distance(g, ORIGIN, min_distance=200, max_distance=100)
