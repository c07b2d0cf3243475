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

import os

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


@pytest.mark.parametrize("detect", [False, True], ids=["index_kernel", "grid_detected"])
@pytest.mark.parametrize("name", list(POSES))
def test_visibility_bit_exact(L, scene, name, detect, monkeypatch):
    """the scene's index array is the full regular grid: with detection (default) the mesh is
    rendered by the LDS-tiled grid kernel, without it by the per-triangle index kernel"""
    if not detect:
        monkeypatch.setenv("ALP_NO_GRID_DETECT", "1")
    p = pose(scene, name)
    ref = orast.visibility(scene["vert"], scene["ind"], p, scene["offsets"])
    with L.Mesh(scene["vert"], None, scene["ind"]) as m:
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        assert_vis_equal(m.fetch_visibility(), ref)
    if name == "looking_away":
        assert not ref.any()
    else:
        assert (ref != 0).mean() > 0.3


@pytest.mark.parametrize("name", ["tilt_roll", "low_near_plane", "inside_looking_north", "tele"])
def test_index_kernel_with_mask_shuffled_array_and_near_plane(L, scene, name, monkeypatch):
    """the per-triangle index kernel on an array no grid detector can use -- shuffled, with a vertex mask, with triangles that
    cross the near plane: the oracle's visibility, triangle ids in the caller's numbering"""
    monkeypatch.setenv("ALP_NO_GRID_DETECT", "1")
    p = dict(scene["params"])
    d = dict(POSES[name] if name in POSES else MORE_POSES[name])
    p["x"] += d.pop("dx", 0.0)
    p["y"] += d.pop("dy", 0.0)
    p["z"] += d.pop("dz", 0.0)
    p.update(d)
    rng = np.random.default_rng(5)
    ind = scene["ind"].astype(np.int32)[rng.permutation(len(scene["ind"]))]
    valid = rng.random(len(scene["vert"])) > 0.03
    keep = valid[ind].all(axis=1)
    ref = orast.visibility(scene["vert"], ind[keep], p, scene["offsets"])
    ids = np.flatnonzero(keep)                  # the oracle numbers the kept triangles 0..K-1, the device the caller's array
    with L.Mesh(scene["vert"], None, ind) as m:
        m.set_valid(valid)
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        got = m.fetch_visibility()
    hit = ref != 0
    np.testing.assert_array_equal(got != 0, hit)
    np.testing.assert_array_equal(got[hit] >> np.uint64(32), ref[hit] >> np.uint64(32))
    tri_dev = 0xFFFFFFFF - (got[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    tri_ref = 0xFFFFFFFF - (ref[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    np.testing.assert_array_equal(tri_dev, ids[tri_ref])


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


MORE_POSES = {
    "inside_looking_north": dict(dx=150.0, dy=-100.0, dz=-30.0, pan=0.0, tilt=-5.0),     # lanes along the other grid axis
    "inside_looking_south_west": dict(dx=200.0, dy=60.0, dz=-35.0, pan=230.0, tilt=-10.0, roll=-15.0),
    "rolled_90": dict(roll=90.0, tilt=-15.0),
    "straight_down": dict(dx=150.0, dz=200.0, tilt=-89.5, pan=10.0),
    "tele": dict(fov=12.0, tilt=-14.0, pan=93.0),                                        # triangles of tens of pixels
}


@pytest.mark.parametrize("name", list(MORE_POSES))
@pytest.mark.parametrize("size", [(640, 427), (1600, 1067)])
def test_more_poses_bit_exact(L, scene, name, size):
    """cameras inside the grid (half of it behind the near plane), other view axes, a large frame
    (triangles above 64 px -> general queue -> work items), on the implicit grid"""
    p = dict(scene["params"])
    d = dict(MORE_POSES[name])
    p["x"] += d.pop("dx", 0.0)
    p["y"] += d.pop("dy", 0.0)
    p["z"] += d.pop("dz", 0.0)
    p.update(d, w=size[0], h=size[1], cx=size[0] / 2, cy=size[1] / 2)
    n = scene["n"]
    ref = orast.visibility(scene["vert"], None, p, scene["offsets"], grid=(n, n))
    with L.Mesh(scene["vert"], None, None, grid=(n, n)) as m:
        m.render_enqueue(L.params_vector(p), scene["offsets"])
        assert_vis_equal(m.fetch_visibility(), ref)
    assert (ref != 0).mean() > 0.05


def test_queue_growth_redoes_the_frame(L, scene, monkeypatch):
    """both device queues start tiny: the first frame overflows them, finish_frame grows them and
    renders again, and the result is the same as with ample queues"""
    p = dict(pose(scene, "low_near_plane"), w=1600, h=1067, cx=800.0, cy=533.5)
    n = scene["n"]
    ref = orast.visibility(scene["vert"], None, p, scene["offsets"], grid=(n, n))
    monkeypatch.setenv("ALP_QUEUE_CAP", "8")
    monkeypatch.setenv("ALP_NO_VIS_CACHE", "1")        # the second frame is drawn again, not served from the first one's visibility
    for kw in (dict(ind=None, grid=(n, n)), dict(ind=scene["ind"].astype(np.int32), grid=None)):
        if kw["ind"] is not None:
            monkeypatch.setenv("ALP_NO_GRID_DETECT", "1")
        with L.Mesh(scene["vert"], None, kw["ind"], grid=kw["grid"]) as m:
            for _ in range(2):                     # second frame: queues already large enough
                m.render_enqueue(L.params_vector(p), scene["offsets"])
                assert_vis_equal(m.fetch_visibility(), ref)


def test_non_square_grid_with_mask(L):
    from alproj_amd import project as prj
    from alproj_amd import synthetic as syn
    s = syn.surface(330)
    gh, gw = 200, 330
    vert = s["vert"][:gh * gw]                      # the 200 northern rows of the 330-column grid
    rng = np.random.default_rng(9)
    valid = rng.random(gh * gw) > 0.02
    a = (np.arange(gw - 1)[None, :] + np.arange(gh - 1)[:, None] * gw).ravel()
    ind = np.stack([a, a + gw, a + gw + 1, a, a + gw + 1, a + 1], axis=1).reshape(-1, 3)
    p = dict(syn.base_params(330), w=640, h=427, cx=320.0, cy=213.5, pan=60.0, tilt=-10.0)
    p["y"] += 60.0
    ref = orast.render(vert, None, ind[valid[ind].all(axis=1)], p, s["offsets"])
    with L.Mesh(vert, None, None, grid=(gh, gw)) as m:
        m.set_valid(valid)
        np.testing.assert_array_equal(prj.persp_proj(m, None, None, p, s["offsets"]), ref)
    assert (ref[:, :, 0] > 0).mean() > 0.1


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
    assert all(df[c].dtype == np.float64 for c in ("x", "y", "z", "B", "G", "R"))
    np.testing.assert_array_equal(df.index.to_numpy(), v * 640 + u)          # labels of the filtered frame
    with pytest.raises(ValueError):
        prj.reverse_proj(sim[:100], scene["vert"], scene["ind"], p, off)          # frame size mismatch
    # no offsets: coordinates stay offset-relative
    from alproj_amd import synthetic as syn
    df0 = prj.reverse_proj(sim, scene["vert"], scene["ind"], syn.local_params(p, off))
    assert len(df0) == len(df) and (df0["x"] > 0).all()
    np.testing.assert_allclose(df0["x"].to_numpy() + off[0], df["x"].to_numpy(), rtol=1e-12)
    with pytest.raises(ValueError):
        prj.reverse_proj(sim, scene["vert"], scene["ind"], p, off, chnames=["a", "b"])


def test_reverse_proj_table_columns_formed_on_the_device(L, scene):
    """to_frame: for uint8 / uint16 / float32 / float64 images every column (labels, u, v, x, y, z, channels) is formed on the
    device (alp_render_fetch_valid_table); other dtypes keep the host gather.  Same frame either way -- values, dtypes, labels."""
    import pandas as pd
    from alproj_amd import project as prj
    p = pose(scene, "tilt_roll")
    off = scene["offsets"]
    rng = np.random.default_rng(9)
    base = rng.integers(0, 60000, (427, 640, 4))
    with prj.reverse_proj_device(scene["vert"], scene["ind"], p, off) as rp:
        want = rp.to_frame(base.astype(np.int32) % 256, list("abcd"))            # int32: the host path
        assert 1000 < len(want) < 427 * 640
        for dt, mod in ((np.uint8, 256), (np.uint16, 60000), (np.float32, 256), (np.float64, 256)):
            img = (base % mod).astype(dt)
            if dt in (np.float32, np.float64):
                img = img + dt(0.37)
            got = rp.to_frame(img, list("abcd"))
            assert list(got.columns) == list(want.columns) and list(got.dtypes) == list(want.dtypes)
            np.testing.assert_array_equal(got.index.to_numpy(), want.index.to_numpy())
            assert got.index.dtype == np.int64 and got["u"].dtype == np.int16 and got["v"].dtype == np.int16
            for c in ("u", "v", "x", "y", "z"):
                np.testing.assert_array_equal(got[c].to_numpy(), want[c].to_numpy())
            flat = img.reshape(-1, 4)[got.index.to_numpy()].astype(np.float64)
            np.testing.assert_array_equal(got[list("abcd")].to_numpy(), flat)
            np.testing.assert_array_equal(got["u"].to_numpy() + 640 * got["v"].to_numpy().astype(np.int64), got.index.to_numpy())
        # a non-contiguous view of a larger image goes through the same entry
        big = rng.integers(0, 256, (427, 640, 6), dtype=np.uint8)
        got = rp.to_frame(big[:, :, ::2], list("abc"))
        np.testing.assert_array_equal(got[list("abc")].to_numpy(), big[:, :, ::2].reshape(-1, 3)[got.index.to_numpy()].astype(np.float64))


def test_set_gcp_on_the_device_equals_the_table_join(L, scene):
    """set_gcp against the resident coordinate image (alp_render_gather) gives
    the rows, labels and values of the reference's merge with the reverse_proj table"""
    import pandas as pd
    from alproj_amd import project as prj
    from alproj_amd.gcp import filter_gcp_distance, set_gcp
    p = pose(scene, "base")
    off = scene["offsets"]
    rng = np.random.default_rng(5)
    n = 2000
    match = pd.DataFrame({"u_org": rng.integers(0, 5616, n), "v_org": rng.integers(0, 3744, n),
                          "u_sim": rng.integers(-3, 643, n), "v_sim": rng.integers(-3, 430, n)})
    dummy = np.zeros((427, 640, 1), np.uint8)
    with prj.reverse_proj_device(scene["vert"], scene["ind"], p, off) as rp:
        got = set_gcp(match, rp)
        frame = rp.to_frame(dummy, ["c"])
        gotf = set_gcp(match.astype(np.float64), rp)
        half = match.astype(np.float64)
        half["u_sim"] += 0.5                      # not a pixel: no row survives
        assert len(set_gcp(half, rp)) == 0
        assert len(set_gcp(match.iloc[:0], rp)) == 0
    exp = set_gcp(match, frame)
    assert 0 < len(exp) < n
    assert list(got.columns) == ["u", "v", "x", "y", "z"]
    np.testing.assert_array_equal(got.index.to_numpy(), exp.index.to_numpy())
    np.testing.assert_array_equal(got.to_numpy(dtype=np.float64), exp.to_numpy(dtype=np.float64))
    np.testing.assert_array_equal(gotf.to_numpy(dtype=np.float64), exp.to_numpy(dtype=np.float64))
    dall = np.sqrt((got["x"] - p["x"]) ** 2 + (got["y"] - p["y"]) ** 2 + (got["z"] - p["z"]) ** 2)
    cut = float(np.median(dall))
    near = filter_gcp_distance(got, p, max_distance=cut)
    assert len(near) == int((dall <= cut).sum()) and 0 < len(near) < len(got)
    assert list(near.index) == list(range(len(near)))


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


@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (1, 3), (2, 2), (7, 5), (64, 1)])
def test_tiny_frames(L, scene, w, h):
    """degenerate image sizes: the centre (w-1)/2 is 0 for a one-pixel axis and the distortion map
    of the reference divides by it (project.py:128-131) -> nothing is drawn; oracle and device agree"""
    from alproj_amd import project as prj
    for extra in ({}, dict(k1=-0.05, a1=1.02)):
        p = dict(pose(scene, "tilt_roll"), w=w, h=h, cx=w / 2, cy=h / 2, **extra)
        ref = orast.render(scene["vert"], scene["col"], scene["ind"], p, scene["offsets"])
        got = prj.persp_proj(scene["vert"], scene["col"], scene["ind"], p, scene["offsets"])
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
        assert (ref.any(axis=2) == got.any(axis=2)).all()


def test_state_and_argument_errors(L, scene):
    """the ABI reports misuse instead of reading garbage"""
    from alproj_amd import project as prj
    with L.Mesh(scene["vert"], None, scene["ind"]) as m:
        m.shape = (427, 640, 3)
        for call in (m.fetch, m.fetch_visibility, m.fetch_valid, lambda: m.gather([1], [1])):
            with pytest.raises(L.AlprojHipError, match="nothing rendered yet"):
                call()
        with pytest.raises(ValueError):
            m.set_valid(np.ones(5))
        with pytest.raises(ValueError):
            m.gather([1, 2], [1])
        m.render_enqueue(L.params_vector(pose(scene, "base")), scene["offsets"], coords=True)
        assert m.gather([], []).shape == (0, 3)
        xyz = m.gather([-1, 640, 3, 320], [5, 5, -2, 400], scene["offsets"])
        assert np.isnan(xyz[:3]).all() and np.isfinite(xyz[3]).all()
    with pytest.raises(L.AlprojHipError, match="image size"):
        prj.persp_proj(scene["vert"], None, scene["ind"], dict(pose(scene, "base"), w=0), scene["offsets"])
    dsm = np.zeros((4, 4), np.float32)
    with pytest.raises(ValueError):
        L.Mesh.from_rasters(dsm, (1, 0, 0, 0, -1, 4), 10.0, np.zeros((3, 4, 5), np.uint8), 255.0)
    with pytest.raises(ValueError):
        L.Mesh.from_rasters(dsm, (1, 0, 0, 0, -1), 10.0, np.zeros((3, 4, 4), np.uint8), 255.0)
    with pytest.raises(L.AlprojHipError, match="z_max is negative"):
        L.Mesh.from_rasters(dsm, (1, 0, 0, 0, -1, 4), -1.0, np.zeros((3, 4, 4), np.uint8), 255.0)
    with pytest.raises(L.AlprojHipError, match="at least 2 x 2"):
        L.Mesh.from_rasters(dsm[:1], (1, 0, 0, 0, -1, 4), 1.0, np.zeros((3, 1, 4), np.uint8), 255.0)


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
    ind32 = syn.grid_indices(n, np.int32)
    # the full regular grid is recognised at mesh creation and rendered by the grid kernel ...
    with L.Mesh(s["vert"], None, ind32) as m:
        m.render_enqueue(pv, s["offsets"])
        assert_vis_equal(m.fetch_visibility(), vis)
    # ... unless told not to: the per-triangle index kernel gives the same frame
    import os
    os.environ["ALP_NO_GRID_DETECT"] = "1"
    try:
        with L.Mesh(s["vert"], None, ind32) as m:
            m.render_enqueue(pv, s["offsets"])
            assert_vis_equal(m.fetch_visibility(), vis)
    finally:
        del os.environ["ALP_NO_GRID_DETECT"]


def _mesh_case(L, vert, ind, p, offsets=None):
    ref = orast.visibility(vert, ind, p, offsets)
    with L.Mesh(vert, None, ind) as m:
        m.render_enqueue(L.params_vector(p), offsets)
        got = m.fetch_visibility()
    assert_vis_equal(got, ref)
    return ref


def test_hand_made_meshes(L):
    """a few large triangles: the 64x64-tile pass, back faces, a shared edge through pixel
    centres, a triangle through the camera plane, and the float64 fallback for projected
    coordinates beyond the fixed-point range"""
    p = dict(x=0.0, y=0.0, z=0.0, fov=60.0, pan=0.0, tilt=0.0, roll=0.0, a1=1, a2=1, k1=0, k2=0, k3=0, k4=0,
             k5=0, k6=0, p1=0, p2=0, s1=0, s2=0, s3=0, s4=0, w=320, h=200, cx=160, cy=100)
    # pan = 0 looks along +Y (north) in X, Z(up), Y storage: view z = stored third component
    quad = np.array([[-20, -8, 60], [20, -8, 60], [20, 8, 60], [-20, 8, 60]], np.float32)
    ind_ccw = np.array([[0, 1, 2], [0, 2, 3]], np.int64)
    r1 = _mesh_case(L, quad, ind_ccw, p)
    r2 = _mesh_case(L, quad, ind_ccw[:, ::-1].copy(), p)
    assert (r1 != 0).any() != (r2 != 0).any()            # one orientation is culled entirely
    front = ind_ccw if (r1 != 0).any() else ind_ccw[:, ::-1].copy()
    vis = r1 if (r1 != 0).any() else r2
    assert 0.2 < (vis != 0).mean() < 0.9
    # both triangles of the quad appear and the diagonal has no hole
    ids = np.unique(0xFFFFFFFF - (vis[vis != 0] & np.uint64(0xFFFFFFFF)).astype(np.int64))
    assert list(ids) == [0, 1]
    rows = np.nonzero((vis != 0).any(axis=1))[0]
    inner = vis[rows[2]:rows[-2], :]
    for row in inner:
        cols = np.nonzero(row)[0]
        assert (np.diff(cols) == 1).all()                # contiguous span: watertight diagonal
    # a ground plane through the camera plane (near clip) seen from 2 units above
    g = 500.0
    ground = np.array([[-g, -2, -g], [g, -2, -g], [g, -2, g], [-g, -2, g]], np.float32)
    _mesh_case(L, ground, front, p)
    _mesh_case(L, ground, front[:, ::-1].copy(), p)
    # a huge triangle whose vertices project millions of pixels away (vz just above 1): float64 fallback
    far = np.array([[-3e6, -2e6, 1.001], [3e6, -2e6, 1.001], [0, 5e6, 2.0]], np.float32)
    for tri in ([[0, 1, 2]], [[0, 2, 1]]):
        _mesh_case(L, far, np.array(tri, np.int64), p)
    # occlusion: a near small quad in front of a far big one, drawn in both orders
    both = np.vstack([quad, quad * np.float32(0.25) + np.float32([0, 0, -30])]).astype(np.float32)
    i2 = np.vstack([front, front + 4])
    a = _mesh_case(L, both, i2, p)
    b = _mesh_case(L, both, i2[::-1].copy(), p)
    centre = a[100, 160]
    assert (centre >> np.uint64(32)) == (b[100, 160] >> np.uint64(32))     # same depth wins regardless of order


def test_frame_size_changes_and_tiny_frames(L, scene):
    from alproj_amd import project as prj
    with prj.Mesh(scene["vert"], scene["col"], scene["ind"]) as m:
        for w, h in ((640, 427), (33, 17), (1, 1), (640, 427)):
            p = dict(pose(scene, "base"), w=w, h=h, cx=w / 2, cy=h / 2)
            got = prj.persp_proj(m, None, None, p, scene["offsets"])
            ref = orast.render(scene["vert"], scene["col"], scene["ind"], p, scene["offsets"])
            assert got.shape == (h, w, 3)
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("corner", ["b_top_right", "d_bottom_left", "a_bottom_right", "c_top_left"])
def test_near_field_cell_poking_a_viewport_corner(L, corner):
    """A grid cell ~600 x 500 px large of which only one corner reaches a viewport corner (at most 2 x 2
    pixel centres).  Its viewport-clamped box is tiny, but its edge vectors are far beyond the 24-bit
    products and the 2^12 tie key of the cell fast path: such a cell must take the general 64-bit path
    (a host emulation of the unguarded fast path got 2 of 9 coverage decisions wrong here)."""
    w, h, fov = 64, 48, 60.0
    p = dict(x=0.0, y=0.0, z=0.0, fov=fov, pan=0.0, tilt=0.0, roll=0.0, a1=1, a2=1, k1=0, k2=0, k3=0, k4=0, k5=0, k6=0,
             p1=0, p2=0, s1=0, s2=0, s3=0, s4=0, w=w, h=h, cx=w / 2, cy=h / 2)
    fx = 1 / np.tan(np.radians(fov) / 2) * (w / 2)
    fy = 1 / np.tan(np.radians(fov) * h / w / 2) * (h / 2)
    depth = 2.0

    def view_xy(xw, yw):                        # window (pixels) -> view-space x, y at `depth`
        return (xw - w / 2) / fx * depth, (yw - h / 2) / fy * depth

    # window rectangle of the cell: x0..x1, y0..y1 (GL window, y up), one corner 2 px inside the viewport
    rect = {"b_top_right": (w - 2.2, w + 598.0, h - 2.3, h + 498.0),        # cell's bottom-left vertex (b) inside
            "d_bottom_left": (-598.0, 2.2, -498.0, 2.3),                   # top-right vertex (d) inside
            "a_bottom_right": (w - 2.2, w + 598.0, -498.0, 2.3),           # top-left vertex (a) inside
            "c_top_left": (-598.0, 2.2, h - 2.3, h + 498.0)}[corner]       # bottom-right vertex (c) inside
    x0, x1, y0, y1 = rect
    (vx0, vy0), (vx1, vy1) = view_xy(x0, y0), view_xy(x1, y1)
    # grid 2 x 2, vertex id = row * 2 + col; row 0 = top (high Z), columns left to right; stored X, Z(up), Y(forward)
    vert = np.array([[vx0, vy1, depth], [vx1, vy1, depth], [vx0, vy0, depth], [vx1, vy0, depth]], dtype=np.float32)
    ref = orast.visibility(vert, None, p, None, grid=(2, 2))
    assert 1 <= (ref != 0).sum() <= 9
    with L.Mesh(vert, None, None, grid=(2, 2)) as m:
        m.render_enqueue(L.params_vector(p), None)
        got = m.fetch_visibility()
    assert_vis_equal(got, ref)
    # and as part of a larger grid whose other cells are ordinary
    from alproj_amd import synthetic as syn
    ind = syn.grid_indices(2, np.int32)
    with L.Mesh(vert, None, ind) as m:
        m.render_enqueue(L.params_vector(p), None)
        assert_vis_equal(m.fetch_visibility(), ref)


@pytest.mark.parametrize("seed", range(6))
def test_random_poses_culling_and_occlusion_stay_exact(L, seed):
    """Random cameras around, above, inside and beside a 400 x 260 surface (non-square: several tile rows and
    columns, partial tiles), small frames so that most cells are far below a pixel: frustum culling, the near /
    far split and the depth-pyramid occlusion test must never change a single visibility word (each frame is
    compared with the frozen oracle and with the same library running with both cullings switched off)."""
    import os
    rng = np.random.default_rng(100 + seed)
    gh, gw = 260, 400
    yy, xx = np.mgrid[0:gh, 0:gw].astype(np.float64)
    z = 60 * np.sin(xx / 45.0) * np.cos(yy / 33.0) + 15 * np.sin(xx / 7.0) * np.sin(yy / 9.0) + rng.normal(0, 0.3, (gh, gw))
    vert = np.stack([xx.ravel(), (z - z.min()).ravel(), (gh - 1 - yy).ravel()], 1).astype(np.float32)      # X, Z(up), Y
    valid = None
    if seed % 2:
        valid = rng.random(gh * gw) > 0.02
    with L.Mesh(vert, None, None, grid=(gh, gw)) as m:
        if valid is not None:
            m.set_valid(valid)
        ind = None
        if valid is not None:          # the oracle draws the filtered index array; ids are mapped below
            from alproj_amd import synthetic as syn
            a = (np.arange(gw - 1)[None, :] + np.arange(gh - 1)[:, None] * gw).ravel()
            full = np.stack([a, a + gw, a + gw + 1, a, a + gw + 1, a + 1], axis=1).reshape(-1, 3)
            keep = np.flatnonzero(valid[full].all(axis=1))
            ind = full[keep]
        for k in range(5):
            w, h = [(160, 120), (97, 64), (333, 200), (64, 200), (243, 90)][k]      # widths off the 8-pixel lines too (LDS patches)
            kind = rng.integers(0, 4)
            if kind == 0:      # beside the surface looking across it
                cam = dict(x=-rng.uniform(5, 300), y=rng.uniform(0, gh), z=float(z.max() - z.min()) + rng.uniform(-40, 80), pan=rng.uniform(60, 120))
            elif kind == 1:    # above it looking down-ish
                cam = dict(x=rng.uniform(0, gw), y=rng.uniform(0, gh), z=float(z.max() - z.min()) + rng.uniform(20, 400), pan=rng.uniform(0, 360))
            elif kind == 2:    # inside the valley system, low above the ground
                cx, cy = rng.integers(5, gw - 5), rng.integers(5, gh - 5)
                cam = dict(x=float(cx), y=float(gh - 1 - cy), z=float(z[cy, cx] - z.min()) + rng.uniform(1.2, 6.0), pan=rng.uniform(0, 360))
            else:              # far away
                cam = dict(x=-rng.uniform(500, 3000), y=rng.uniform(-500, gh + 500), z=rng.uniform(50, 800), pan=rng.uniform(70, 110))
            p = dict(orc.vector_to_params(np.zeros(25)), a1=1.0, a2=1.0, fov=rng.uniform(25, 88), tilt=rng.uniform(-60, 15),
                     roll=rng.uniform(-25, 25), w=w, h=h, cx=w / 2, cy=h / 2, **cam)
            pv = L.params_vector(p)
            m.render_enqueue(pv, None)
            got = m.fetch_visibility()
            ref = orast.visibility(vert, ind, p, None, grid=None if ind is not None else (gh, gw))
            if ind is None:
                assert_vis_equal(got, ref)
            else:
                hit = ref != 0
                np.testing.assert_array_equal(got != 0, hit)
                np.testing.assert_array_equal(got[hit] >> np.uint64(32), ref[hit] >> np.uint64(32))
                tri_dev = 0xFFFFFFFF - (got[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
                tri_ref = 0xFFFFFFFF - (ref[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
                np.testing.assert_array_equal(tri_dev, keep[tri_ref])
            os.environ["ALP_NO_TILE_CULL"] = "1"
            os.environ["ALP_NO_VIS_CACHE"] = "1"          # same view again: really draw it again
            try:
                full = m.frame_counts()[0]
                m.render_enqueue(pv, None)
                assert m.frame_counts()[0] == full + 1
                np.testing.assert_array_equal(m.fetch_visibility(), got)
            finally:
                os.environ.pop("ALP_NO_TILE_CULL", None)
                os.environ.pop("ALP_NO_VIS_CACHE", None)


@pytest.mark.parametrize("seed", range(8))
def test_mid_field_frames_with_depth_patches_stay_exact(L, seed):
    """Frames in which a cell is 0.5 ... 8 pixels and a tile of 64 x 16 cells has a footprint of 1 000 ... 30 000
    pixels: the first-round tiles that collect their fragments in an LDS depth patch, those that do not fit one,
    parked cells and parked triangles next to each other.  Every visibility word against the frozen oracle,
    with the patches on, and the same frame with the culling switched off."""
    import os
    rng = np.random.default_rng(500 + seed)
    gh, gw = 420, 610
    yy, xx = np.mgrid[0:gh, 0:gw].astype(np.float64)
    z = 40 * np.sin(xx / 70.0 + seed) * np.cos(yy / 55.0) + 6 * np.sin(xx / 5.0) * np.sin(yy / 6.0) + rng.normal(0, 0.25, (gh, gw))
    vert = np.stack([xx.ravel(), (z - z.min()).ravel(), (gh - 1 - yy).ravel()], 1).astype(np.float32)      # X, Z(up), Y
    w, h = [(640, 427), (901, 600), (1111, 733), (512, 512), (777, 333), (640, 427), (1280, 200), (333, 777)][seed]
    side = seed % 4
    height = float(z.max() - z.min())
    if side == 0:
        cam = dict(x=-rng.uniform(20, 200), y=rng.uniform(0.2, 0.8) * gh, z=height + rng.uniform(5, 120), pan=rng.uniform(75, 105))
    elif side == 1:
        cam = dict(x=gw + rng.uniform(20, 200), y=rng.uniform(0.2, 0.8) * gh, z=height + rng.uniform(5, 120), pan=rng.uniform(255, 285))
    elif side == 2:
        cam = dict(x=rng.uniform(0.2, 0.8) * gw, y=-rng.uniform(20, 200), z=height + rng.uniform(5, 120), pan=rng.uniform(-15, 15) % 360)
    else:
        cam = dict(x=rng.uniform(0.3, 0.7) * gw, y=rng.uniform(0.3, 0.7) * gh, z=height + rng.uniform(60, 250), pan=rng.uniform(0, 360))
    p = dict(orc.vector_to_params(np.zeros(25)), a1=1.0, a2=1.0, fov=rng.uniform(35, 75), tilt=rng.uniform(-50, -8) if side < 3 else rng.uniform(-89, -60),
             roll=rng.uniform(-10, 10), w=w, h=h, cx=w / 2, cy=h / 2, **cam)
    valid = rng.random(gh * gw) > 0.01 if seed % 3 == 0 else None
    a = (np.arange(gw - 1)[None, :] + np.arange(gh - 1)[:, None] * gw).ravel()
    full = np.stack([a, a + gw, a + gw + 1, a, a + gw + 1, a + 1], axis=1).reshape(-1, 3)
    keep = np.flatnonzero(valid[full].all(axis=1)) if valid is not None else None
    ref = orast.visibility(vert, full if keep is None else full[keep], p, None)
    assert (ref != 0).mean() > 0.03         # the camera sees the surface (a sanity check of the case, not of the kernel)
    with L.Mesh(vert, None, None, grid=(gh, gw)) as m:
        if valid is not None:
            m.set_valid(valid)
        pv = L.params_vector(p)
        m.render_enqueue(pv, None)
        got = m.fetch_visibility()
        if keep is None:
            assert_vis_equal(got, ref)
        else:
            hit = ref != 0
            np.testing.assert_array_equal(got != 0, hit)
            np.testing.assert_array_equal(got[hit] >> np.uint64(32), ref[hit] >> np.uint64(32))
            np.testing.assert_array_equal(0xFFFFFFFF - (got[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64),
                                          keep[0xFFFFFFFF - (ref[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)])
        for env in ("ALP_NO_TILE_CULL", "ALP_NO_OCCLUSION"):
            os.environ[env] = "1"
            os.environ["ALP_NO_VIS_CACHE"] = "1"          # same view again: really draw it again
            try:
                full = m.frame_counts()[0]
                m.render_enqueue(pv, None)
                assert m.frame_counts()[0] == full + 1
                np.testing.assert_array_equal(m.fetch_visibility(), got)
            finally:
                os.environ.pop(env, None)
                os.environ.pop("ALP_NO_VIS_CACHE", None)


@pytest.mark.parametrize("pan", [95.0, 200.0, 318.0])
def test_unoffset_utm_scale_coordinates_stay_exact(L, pan):
    """vertices and camera at UTM magnitude WITHOUT offsets (float32 ulp 0.06 ... 0.5 m): the tile culling's and the
    occlusion test's margins carry an absolute term for the rounding of the float32 box centres and camera position,
    so the culled frame still equals the frozen oracle's and the frame drawn without culling"""
    from alproj_amd import synthetic as syn
    n = 700
    s = syn.surface(n, res=4.0)
    vert = s["vert"].astype(np.float64)
    vert[:, 0] += 732000.0
    vert[:, 2] += 4048000.0
    vert = vert.astype(np.float32)                          # quantised to the float32 grid at that magnitude
    p = dict(syn.base_params(n, 4.0), w=800, h=533, cx=400.0, cy=266.5, pan=pan, tilt=-6.0)
    p.update(x=float(vert[n * (n // 2) + n // 3, 0]), y=float(vert[n * (n // 2) + n // 3, 2]), z=float(vert[n * (n // 2) + n // 3, 1]) + 35.0)
    ref = orast.visibility(vert, None, p, None, grid=(n, n))
    assert (ref != 0).mean() > 0.2
    with L.Mesh(vert, None, None, grid=(n, n)) as m:
        m.render_enqueue(L.params_vector(p), None)
        got = m.fetch_visibility()
        assert_vis_equal(got, ref)
        for env in ("ALP_NO_TILE_CULL", "ALP_NO_OCCLUSION"):
            os.environ[env] = "1"
            os.environ["ALP_NO_VIS_CACHE"] = "1"
            try:
                m.render_enqueue(L.params_vector(p), None)
                np.testing.assert_array_equal(m.fetch_visibility(), got)
            finally:
                os.environ.pop(env, None)
                os.environ.pop("ALP_NO_VIS_CACHE", None)
