"""GPU: the render wrappers, the distortion map and LsqOptimizer against vectors captured from the
reference's own functions (tests/golden/gen_golden_render.py: g12, g13, g14), through the C ABI."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import ref_numpy as orc
from tests.test_oracle_golden_render import COEFFS, SIZES, check_frame

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


@pytest.mark.parametrize("otag", ["off", "nooff"])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g12_reverse_proj_post_processing_on_the_device(L, tag, otag):
    """the reference's reverse_proj (project.py:361-373) ran unmodified on a seeded raw render; the same
    raw image installed as the device frame must give the same DataFrame: x > 0 selection incl.
    0 / negative / tiny / NaN / inf, channel reorder, int16 u, v, offsets, row labels."""
    from alproj_amd import project as aproj
    g = load("g12_wrappers.npz")
    raw, array = g[f"{tag}_raw"], g[f"{tag}_array"]
    off = g["offsets"] if otag == "off" else None
    h, w = raw.shape[:2]
    vert = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0, 1]], dtype=np.float32)      # any mesh: the frame is loaded
    with L.Mesh(vert, None, None, grid=(2, 2)) as m:
        m.load_image(raw)
        rp = aproj.ReverseProjection(m, off, w, h, False, None)
        check_frame(rp.to_frame(array, list(g[f"{tag}_chnames"])), g, tag, otag)
        # set_gcp's gather on the same frame = a lookup in the reference's table
        df = pd.DataFrame(g[f"{tag}_{otag}_values"], columns=list(g[f"{tag}_{otag}_columns"]))
        pick = df.iloc[:: max(1, len(df) // 50)]
        got = rp.lookup(pick["u"].to_numpy(), pick["v"].to_numpy())
        np.testing.assert_array_equal(got, pick[["x", "y", "z"]].to_numpy())
        assert np.isnan(rp.lookup([0], [0])).all()            # raw[0, 0] has x == 0: not in the table


def test_g12_sim_image_tail_on_the_device(L):
    """the reference's sim_image ran unmodified on a seeded colour render (g12): alp_render_fetch_u8 on the same
    frame returns the same bytes; values outside [0, 1] (never produced by get_colored_surface, surface.py:66)
    wrap like numpy's astype(uint8) on x86-64"""
    g = load("g12_wrappers.npz")
    raw = g["sim_raw"]
    vert = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0, 1]], dtype=np.float32)
    with L.Mesh(vert, None, None, grid=(2, 2)) as m:
        m.load_image(raw)
        out = m.fetch_u8(255.0, True)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
        np.testing.assert_array_equal(out, g["sim_bgr"])
        wild = np.array([[[-0.01, 1.004, 2.5], [300.7, -1.2, 0.5], [np.nan, 1e12, -1e12]]], dtype=np.float32)
        m.load_image(wild)
        x = wild * np.float32(255)
        ok = np.isfinite(x) & (np.abs(x) < 2.0**31)
        expect = np.where(ok, np.trunc(np.where(ok, x, 0)).astype(np.int64) & 0xFF, 0).astype(np.uint8)
        np.testing.assert_array_equal(m.fetch_u8(255.0, False), expect)


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("name", COEFFS)
def test_g13_distort_map_on_the_device(L, name, size):
    """alp_distort_map = the float32 map_x / map_y the reference's distort() handed to cv2.remap, bit
    for bit; alp_distort_image gathers from its round-half-even (constant-0 border)."""
    g = load("g13_distort_map.npz")
    h, w = (int(s) for s in size.split("x"))
    mx, my = L.distort_map(h, w, g[f"coeffs_{name}"])
    np.testing.assert_array_equal(mx, g[f"mapx_{name}_{size}"])
    np.testing.assert_array_equal(my, g[f"mapy_{name}_{size}"])
    index_img = (np.arange(h * w, dtype=np.float32) + 1).reshape(h, w, 1)
    got = L.distort_image(index_img, g[f"coeffs_{name}"])
    np.testing.assert_array_equal(got, orc.remap_nearest(index_img, g[f"mapx_{name}_{size}"], g[f"mapy_{name}_{size}"]))


LSQ_KW = {"trf_linear_d7": dict(method="trf"),
          "trf_huber_d7": dict(method="trf", loss="huber", f_scale=5.0),
          "dogbox_softl1_d4": dict(method="dogbox", loss="soft_l1", f_scale=3.0,
                                   bound_widths={"fov": 10, "pan": 10, "tilt": 10, "roll": 10}),
          "trf_cauchy_d4": dict(method="trf", loss="cauchy", f_scale=2.0),
          "lm_d4": dict(method="lm"),
          "trf_linear_dist_d6": dict(method="trf")}


@pytest.mark.parametrize("case", list(LSQ_KW))
def test_g14_lsq_optimizer_matches_the_reference_run(L, case):
    """LsqOptimizer.optimize of the reference (optimize.py:467-539, scipy least_squares) on seeded,
    well-posed GCP problems: same optimum and error from the device residuals.  jac='2-point' =
    scipy's own sequential differences (the reference's call); 'batched' = the D+1 poses of the same
    scheme in one launch.  Tolerance: the reference's own optimum moves when its residuals are
    perturbed by 1e-13 (finite differences with a 1.5e-8 step amplify them; least_squares stops at
    ftol = 1e-8), see below."""
    from alproj_amd import optimize as aopt
    g = load("g14_lsq.npz")
    keys = [str(k) for k in g["param_keys"]]
    init = dict(zip(keys, g[f"{case}_init"]))
    dfx = pd.DataFrame(g["xyz"], columns=["x", "y", "z"])
    dfu = pd.DataFrame(g["uv_" + str(g[f"{case}_uv"])], columns=["u", "v"])
    targets = [str(t) for t in g[f"{case}_targets"]]
    want = dict(zip(keys, g[f"{case}_params"]))
    for jac in ("2-point", "batched"):
        o = aopt.LsqOptimizer(dfx, dfu, dict(init))
        o.set_target(targets)
        params, err = o.optimize(jac=jac, **LSQ_KW[case])
        assert set(params) == set(want)
        # measured on the reference itself (1e-13 .. 1e-12 residual perturbations): pose parameters move
        # by up to 2e-5 (metres / degrees), distortion coefficients by 2e-7, the error by 1e-5 relative
        for k in targets:
            tol = 2e-4 if k in ("x", "y", "z", "fov", "pan", "tilt", "roll") else 2e-6
            assert abs(params[k] - want[k]) <= tol, (jac, k, params[k], want[k])
        assert err == pytest.approx(float(g[f"{case}_error"]), rel=5e-5)
        assert err < 2.5


@pytest.mark.parametrize("tag", ["a", "c"])
def test_g12_table_free_rasterize_equals_rasterizing_the_references_table(L, tag):
    """reverse_proj -> to_geotiff without the table (ReverseProjection.rasterize) against the rasterisation of the table the
    REFERENCE's reverse_proj returned for the same frame (g12): same bytes, same bounds, for every aggregate"""
    from alproj_amd import project as aproj
    g = load("g12_wrappers.npz")
    raw, array = g[f"{tag}_raw"], g[f"{tag}_array"]
    chn = list(g[f"{tag}_chnames"])
    h, w = raw.shape[:2]
    ref_table = pd.DataFrame(g[f"{tag}_off_values"], columns=list(g[f"{tag}_off_columns"]), index=g[f"{tag}_off_index"])
    # g12 plants x = inf at pixel (1, 1) for the x > 0 filter; a raster has no extent for it (the reference's to_geotiff
    # overflows on it, and so do both paths here): that one coordinate becomes finite in the frame AND in the reference's row
    assert np.isinf(raw[1, 1, 0]) and np.isinf(ref_table.loc[w + 1, "x"])
    raw = raw.copy()
    raw[1, 1, 0] = np.float32(77.0)
    ref_table.loc[w + 1, "x"] = 77.0 + g["offsets"][0]
    with pytest.raises(OverflowError):
        aproj.rasterize(pd.DataFrame(g[f"{tag}_off_values"], columns=list(g[f"{tag}_off_columns"])), resolution=40.0, bands=chn[:1])
    vert = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0, 1]], dtype=np.float32)
    with L.Mesh(vert, None, None, grid=(2, 2)) as m:
        m.load_image(raw)
        rp = aproj.ReverseProjection(m, g["offsets"], w, h, False, None)
        for agg in ("mean", "median", "max", "min"):
            bands = chn[:3][::-1]
            want, wb = aproj.rasterize(ref_table, resolution=40.0, bands=bands, max_dist=80.0, agg_func=agg)
            got, gb = rp.rasterize(array, chn, resolution=40.0, bands=bands, max_dist=80.0, agg_func=agg)
            assert gb == wb
            np.testing.assert_array_equal(got, want)
