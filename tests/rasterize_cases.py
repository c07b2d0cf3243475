"""Seeded inputs shared by tests/golden/gen_golden_geotiff_float.py (the reference's to_geotiff in the build container) and
tests/test_gpu_rasterize.py / test_oracle_geotiff.py: a million FLOAT-valued points (the fixture stores only what the
reference wrote, the inputs are rebuilt from the seed)."""
import numpy as np
import pandas as pd


def float_points(n=1_000_000, seed=20261004, span_x=520.0, span_y=410.0):
    """points the way a camera delivers them: clustered (hundreds share a cell next to the 'camera', one or none far away),
    band values with fractions of all magnitudes -- R in [0, 256) like a colour with noise, G spanning 1e-3 ... 1e3 so that sums
    of a cell mix exponents (where the ORDER of a float64 sum shows), B near the byte boundaries k + 1 - 2^-k -- and NaN in 0.1 %
    of one band"""
    rng = np.random.default_rng(seed)
    u = rng.random(n) ** 2.2                                   # dense towards x = 0
    x = 732000.0 + u * span_x
    y = 4048000.0 + rng.random(n) * span_y * (0.15 + 0.85 * u)
    keep = ~((u > 0.55) & (u < 0.6))                           # a band without points: NaN cells for the focal fill
    x, y = x[keep], y[keep]
    m = len(x)
    r = rng.integers(0, 256, m) + rng.random(m)
    g = 10.0 ** rng.uniform(-3, 2.4, m)
    k = rng.integers(0, 255, m)
    b = k + 1.0 - 2.0 ** -rng.integers(1, 40, m)
    g[rng.random(m) < 1e-3] = np.nan
    return pd.DataFrame({"x": x, "y": y, "R": np.minimum(r, 255.999), "G": g, "B": b})


FLOAT_CASES = {
    "float_mean": dict(resolution=1.0, agg_func="mean", max_dist=1.0),
    "float_median": dict(resolution=1.0, agg_func="median", max_dist=1.0),
    "float_max_res2": dict(resolution=2.0, agg_func="max", max_dist=2.0),
}


def byte_points(n=1_000_000, seed=20261005, span_x=520.0, span_y=410.0):
    """the reference's own use of to_geotiff: the same clustered points carrying a photograph's bytes (uint8 R, G, B in float64
    columns, project.py:364) -- runs of 1, 2, 3-16 and of hundreds of points per cell, the cases the packed kernels tell apart"""
    rng = np.random.default_rng(seed)
    u = rng.random(n) ** 2.2
    x = 732000.0 + u * span_x
    y = 4048000.0 + rng.random(n) * span_y * (0.15 + 0.85 * u)
    keep = ~((u > 0.55) & (u < 0.6))
    x, y = x[keep], y[keep]
    m = len(x)
    base = (128 + 100 * np.sin(x / 37.0) * np.cos(y / 23.0)).astype(np.int64)      # a picture, not noise: neighbouring points agree
    cols = {c: np.clip(base + rng.integers(-40, 41, m) + d, 0, 255).astype(np.float64) for c, d in (("R", 10), ("G", 0), ("B", -25))}
    return pd.DataFrame({"x": x, "y": y, **cols})


BYTE_CASES = {
    "byte_mean": dict(resolution=1.0, agg_func="mean", max_dist=1.0),
    "byte_median": dict(resolution=1.0, agg_func="median", max_dist=1.0),
    "byte_max_res2": dict(resolution=2.0, agg_func="max", max_dist=2.0),
    "byte_min_gb": dict(resolution=1.0, agg_func="min", max_dist=3.0, bands=["G", "B"], nodata=0),
    "byte_median_r_half": dict(resolution=0.5, agg_func="median", interpolate=False, bands=["R"]),
}
