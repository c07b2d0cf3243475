"""The oracle's restatement of to_geotiff's compute part against the rasters the reference
itself handed to its GeoTIFF writer (tests/golden/gen_golden_geotiff.py).  CPU only."""
import ast
import os
import warnings

import numpy as np
import pytest

from oracle import ref_numpy as orc

G = os.path.join(os.path.dirname(__file__), "golden", "g9_geotiff.npz")
CASES = ["mean_int", "mean_float_res2", "max_nointerp", "min_sparse", "median_small"]


@pytest.mark.parametrize("name", CASES)
def test_rasterize_matches_reference(name):
    g = np.load(G, allow_pickle=False)
    kw = ast.literal_eval(str(g[f"{name}_kw"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")           # nanmean of all-NaN windows warns in the reference too
        raster, bounds = orc.rasterize_points(g[f"{name}_x"], g[f"{name}_y"], g[f"{name}_vals"], **kw)
    np.testing.assert_array_equal(raster, g[f"{name}_raster"])
    assert (raster.shape[1], raster.shape[2]) == tuple(g[f"{name}_hw"])
    np.testing.assert_array_equal(np.array(bounds[:4]), g[f"{name}_bounds"])


def test_oracle_matches_reference_on_a_million_float_points():
    """g17 (gen_golden_geotiff_float.py): the reference's to_geotiff on a million clustered float-valued points; the oracle's
    mean raster byte for byte (the other aggregates of the fixture are held against the device path, tests/test_gpu_rasterize.py)"""
    from tests.rasterize_cases import FLOAT_CASES, float_points
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g17_geotiff_float.npz"), allow_pickle=False)
    df = float_points()
    assert len(df) == int(g["n_points"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        raster, bounds = orc.rasterize_points(df["x"].to_numpy(), df["y"].to_numpy(), df[["R", "G", "B"]].to_numpy(), **FLOAT_CASES["float_mean"])
    np.testing.assert_array_equal(raster, g["float_mean_raster"])
    np.testing.assert_array_equal(np.array(bounds[:4]), g["float_mean_bounds"])


def test_oracle_matches_reference_on_a_million_byte_points():
    """g18 (gen_golden_geotiff_bytes.py): the reference's to_geotiff on a million clustered byte-valued points -- its own use, a
    photograph's uint8 bands; the oracle's minimum of two bands with three sweeps and nodata 0 byte for byte (the whole fixture is held against
    the device path, tests/test_gpu_rasterize.py)"""
    from tests.rasterize_cases import BYTE_CASES, byte_points
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g18_geotiff_bytes.npz"), allow_pickle=False)
    df = byte_points()
    assert len(df) == int(g["n_points"])
    kw = dict(BYTE_CASES["byte_min_gb"])
    bands = kw.pop("bands")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        raster, bounds = orc.rasterize_points(df["x"].to_numpy(), df["y"].to_numpy(), df[bands].to_numpy(), kw["resolution"], True,
                                              kw["max_dist"], kw["agg_func"], kw["nodata"])
    np.testing.assert_array_equal(raster, g["byte_min_gb_raster"])
    np.testing.assert_array_equal(np.array(bounds[:4]), g["byte_min_gb_bounds"])
