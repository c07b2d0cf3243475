"""GPU parity tests of the render path (persp_proj / sim_image / reverse_proj / distort)
through the C ABI, against the C raster oracle (oracle/raster_ref.c).

The oracle is this project's own restatement of what the reference asks OpenGL / cv2 to do:
PARITY UNPINNED against a real GL driver (moderngl, glcontext and cv2 are not installed and
the reference has no fixture for the render).  What IS checked:
  * visibility (which triangle wins which pixel, and its depth bits): integer/index work,
    bit-exact against the oracle;
  * image values: float64 interpolation on both sides, 1e-6 relative (+1e-6 absolute);
  * the wrappers' semantics (flip, channel order, uint8 cast, x > 0 filter, offsets).
"""
import warnings

import numpy as np
import pytest

from oracle import raster as orast
from oracle import ref_numpy as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


@pytest.fixture(scope="module")
def scene():
    from alproj_amd import synthetic as syn
    n = 300
    s = syn.surface(n)
    p = dict(syn.base_params(n), w=640, h=427, cx=320.0, cy=213.5)
    return dict(n=n, vert=s["vert"], offsets=s["offsets"], ind=syn.grid_indices(n), params=p,
                col=syn.colors(n * n))


POSES = {
    "base": {},
    "tilt_roll": dict(tilt=-12.0, roll=7.0, pan=80.0),
    "low_near_plane": dict(dz=-48.5, tilt=-20.0),        # 1.5 m above ground: triangles cross vz = 1
    "wide": dict(fov=88.0, tilt=-30.0, pan=120.0),
    "looking_away": dict(pan=275.0),                     # surface behind the camera -> empty
}


def pose(scene, name):
    p = dict(scene["params"])
    d = dict(POSES[name])
    p["z"] += d.pop("dz", 0.0)
    p.update(d)
    return p


def assert_vis_equal(got, ref):
    bad = got != ref
    assert not bad.any(), (f"{bad.sum()} of {bad.size} pixels differ; first at {np.argwhere(bad)[0]}: "
                           f"{got[bad][0]:#x} vs {ref[bad][0]:#x}")


@pytest.mark.parametrize("name", list(POSES))
def test_visibility_bit_exact(L, scene, name):
    p = pose(scene, name)
    ref = orast.visibility(scene["vert"], scene["ind"], p, scene["offsets"])
    with L.Mesh(scene["vert"], None, scene["ind"]) as m:
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        assert_vis_equal(m.fetch_visibility(), ref)
    if name == "looking_away":
        assert not ref.any()
    else:
        assert (ref != 0).mean() > 0.3


def test_implicit_grid_and_int32_indices(L, scene):
    p = pose(scene, "tilt_roll")
    n = scene["n"]
    ref = orast.visibility(scene["vert"], scene["ind"], p, scene["offsets"])
    with L.Mesh(scene["vert"], None, None, grid=(n, n)) as m:
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        assert_vis_equal(m.fetch_visibility(), ref)
    with L.Mesh(scene["vert"], None, scene["ind"].astype(np.int32)) as m:
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        assert_vis_equal(m.fetch_visibility(), ref)


def test_filtered_triangles(L, scene):
    """nodata triangles removed from the index array (surface.py:203-205) leave holes"""
    rng = np.random.default_rng(3)
    keep = rng.random(len(scene["ind"])) > 0.3
    ind = scene["ind"][keep]
    p = pose(scene, "base")
    ref = orast.visibility(scene["vert"], ind, p, scene["offsets"])
    with L.Mesh(scene["vert"], None, ind) as m:
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        assert_vis_equal(m.fetch_visibility(), ref)


@pytest.mark.parametrize("name,dist,mind", [("base", {}, None), ("tilt_roll", dict(k1=-0.12, k2=0.02, a1=1.04, p1=0.01, s3=-0.004), 80.0),
                                            ("low_near_plane", dict(k4=0.05, p2=-0.01), 5.0)])
def test_persp_proj_image(L, scene, name, dist, mind):
    from alproj_amd import project as prj
    p = dict(pose(scene, name), **dist)
    ref = orast.render(scene["vert"], scene["col"], scene["ind"], p, scene["offsets"], mind)
    got = prj.persp_proj(scene["vert"], scene["col"], scene["ind"], p, scene["offsets"], mind)
    assert got.shape == (427, 640, 3) and got.dtype == np.float32
    assert (ref.any(axis=2) == got.any(axis=2)).all()
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
    if mind:
        assert (ref[-40:].sum(axis=2) == 0).mean() > 0.2        # near field masked to black


def test_reverse_proj_and_sim_image(L, scene):
    from alproj_amd import project as prj
    p = pose(scene, "base")
    off = scene["offsets"]
    sim = prj.sim_image(scene["vert"], scene["col"], scene["ind"], p, off, min_distance=30)
    raw = orast.render(scene["vert"], scene["col"], scene["ind"], p, off, 30)
    exp = (raw * 255).astype(np.uint8)[:, :, ::-1]
    assert sim.dtype == np.uint8 and sim.shape == (427, 640, 3)
    assert (np.abs(sim.astype(int) - exp.astype(int)) <= 1).all() and (sim == exp).mean() > 0.999
    df = prj.reverse_proj(sim, scene["vert"], scene["ind"], p, off)
    assert list(df.columns) == ["u", "v", "x", "y", "z", "B", "G", "R"]
    assert df["u"].dtype == np.int16 and df["v"].dtype == np.int16
    coord = orast.render(scene["vert"], None, scene["ind"], p, off)
    seen = coord[:, :, 0] > 0
    assert len(df) == int(seen.sum())
    v, u = np.nonzero(seen)
    np.testing.assert_array_equal(df["u"].to_numpy(), u)
    np.testing.assert_array_equal(df["v"].to_numpy(), v)
    # X,Z,Y storage -> x, y, z columns, offsets added back (project.py:361, :370-373)
    np.testing.assert_allclose(df["x"].to_numpy(), coord[seen][:, 0] + off[0], rtol=1e-9)
    np.testing.assert_allclose(df["y"].to_numpy(), coord[seen][:, 2] + off[2], rtol=1e-9)
    np.testing.assert_allclose(df["z"].to_numpy(), coord[seen][:, 1] + off[1], rtol=1e-9)
    np.testing.assert_array_equal(df["B"].to_numpy(), sim[seen][:, 0])
    with pytest.raises(ValueError):
        prj.reverse_proj(sim, scene["vert"], scene["ind"], p, off, chnames=["a", "b"])


def test_reverse_proj_hits_the_surface(L, scene):
    """geometry check independent of the oracle: a pixel's reverse-projected world point must
    project back onto that pixel through the GL camera model"""
    from alproj_amd import project as prj
    p = pose(scene, "tilt_roll")
    off = scene["offsets"]
    dummy = np.zeros((427, 640, 1), np.uint8)
    df = prj.reverse_proj(dummy, scene["vert"], None, p, off, chnames=["c"], grid_shape=(scene["n"], scene["n"]))
    sub = df.iloc[::97]
    mv = orc.modelview_mat(p["pan"], p["tilt"], p["roll"], p["x"], p["y"], p["z"]).reshape(4, 4).T
    pts = np.stack([sub["x"], sub["z"], sub["y"], np.ones(len(sub))])           # X, Z, Y order
    view = mv @ pts
    fx = 1 / np.tan(np.radians(p["fov"]) / 2)
    fy = 1 / np.tan(np.radians(p["fov"]) * p["h"] / p["w"] / 2)
    xw = (fx * view[0] / view[2] + 1) * p["w"] / 2
    yw = (fy * view[1] / view[2] + 1) * p["h"] / 2
    np.testing.assert_allclose(xw, sub["u"] + 0.5, atol=2e-2)
    np.testing.assert_allclose(p["h"] - yw, sub["v"] + 0.5, atol=2e-2)


def test_distort_image(L):
    from alproj_amd import project as prj
    rng = np.random.default_rng(0)
    img = rng.random((120, 200, 3)).astype(np.float32)
    ident = [1, 1] + [0] * 12
    np.testing.assert_array_equal(prj.distort(img, np.array(ident, float)), img)
    coeffs = np.array([1.05, 0.97, -0.1, 0.02, 0.001, 0.01, 0, 0, 0.01, -0.01, 0.003, 0, -0.002, 0.001])
    got = prj.distort(img, coeffs)
    np.testing.assert_array_equal(got, orast.distort_image(img, coeffs))
    np.testing.assert_array_equal(got, orc.distort_image(img, coeffs))        # numpy restatement agrees
    u8 = (img * 255).astype(np.uint8)
    assert prj.distort(u8, coeffs).dtype == np.uint8


def test_edge_cases(L, scene):
    from alproj_amd import project as prj
    p = pose(scene, "base")
    empty = np.zeros((0, 3), dtype=np.int64)
    out = prj.persp_proj(scene["vert"], scene["col"], empty, p, scene["offsets"])
    assert out.shape == (427, 640, 3) and not out.any()
    bad = scene["ind"][:10].copy()
    bad[3, 1] = scene["n"] ** 2          # one past the last vertex
    with pytest.raises(L.AlprojHipError):
        prj.persp_proj(scene["vert"], scene["col"], bad, p, scene["offsets"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        prj.persp_proj(scene["vert"], scene["col"], scene["ind"][:100], dict(p, fov=100.0), scene["offsets"])
        assert any("Wider FoV" in str(x.message) for x in w)
    # a device-resident mesh re-rendered with another pose
    with prj.Mesh(scene["vert"], scene["col"], scene["ind"]) as m:
        a = prj.persp_proj(m, None, None, p, scene["offsets"])
        b = prj.persp_proj(m, None, None, pose(scene, "tilt_roll"), scene["offsets"])
        a2 = prj.persp_proj(m, None, None, p, scene["offsets"])
    np.testing.assert_array_equal(a, a2)                     # idempotent
    assert not np.array_equal(a, b)


def test_dsm_10m_full_frame(L):
    """BASELINE config-4 style render at 10 M vertices / 20 M triangles onto the 5616x3744
    frame: bit-exact visibility against the oracle, implicit grid == explicit int32 indices."""
    from alproj_amd import synthetic as syn
    n = syn.grid_side(10_000_000)
    s = syn.surface(n)
    p = syn.base_params(n)
    pv = L.params_vector(p)
    ref = orast.visibility(s["vert"], None, p, s["offsets"], grid=(n, n))
    with L.Mesh(s["vert"], None, None, grid=(n, n)) as m:
        m.render_enqueue(pv, s["offsets"])
        vis = m.fetch_visibility()
    assert_vis_equal(vis, ref)
    assert (ref != 0).mean() > 0.5
    with L.Mesh(s["vert"], None, syn.grid_indices(n, np.int32)) as m:
        m.render_enqueue(pv, s["offsets"])
        assert_vis_equal(m.fetch_visibility(), vis)
