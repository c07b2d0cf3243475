"""mesh construction on the device (alp_mesh_from_rasters) against the
arrays of the reference's own get_colored_surface, and renders of the masked implicit grid
against the oracle run on the reference's filtered index array."""
import os
import warnings

import numpy as np
import pytest

from alproj_amd import synthetic as syn
from oracle import raster as orast

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_surface.npz"))
NAMES = [str(n) for n in G["names"]]


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


def camera(vert, offsets, w=320, h=200):
    """a camera above one corner of the (offset-relative, X Z Y) patch looking across it"""
    p = dict(syn.BASE_CAMERA)
    ext = vert.max(axis=0)
    p.update(x=offsets[0] - 5.0, y=offsets[2] + ext[2] / 2, z=offsets[1] + ext[1] + 15.0, pan=90.0, tilt=-25.0,
             fov=70.0, w=w, h=h, cx=w / 2, cy=h / 2, k1=-0.03, p1=1e-3)
    return p


@pytest.mark.parametrize("name", NAMES)
def test_device_mesh_equals_reference_arrays(L, name):
    from alproj_amd import project as prj
    from alproj_amd.surface import colored_surface_mesh
    cm = float(G[f"{name}_color_max"])
    aerial, nodata = G[f"{name}_aerial"], G[f"{name}_nodata"]
    mesh, off = colored_surface_mesh(aerial, G[f"{name}_filled"], G[f"{name}_transform"], nodata, aerial.dtype,
                                     color_max=None if np.isnan(cm) else cm,
                                     dsm_max_height=np.float32(G[f"{name}_zmax"]))
    with mesh:
        vert, col, valid = mesh.fetch_arrays()
        np.testing.assert_array_equal(off, G[f"{name}_offsets"])
        np.testing.assert_array_equal(vert, G[f"{name}_vert"].astype(np.float32))      # persp_proj's cast, project.py:213
        np.testing.assert_array_equal(col, G[f"{name}_col"].astype(np.float32))
        np.testing.assert_array_equal(valid, ~nodata.ravel())
        # the masked implicit grid renders what the reference's filtered index array renders
        p = camera(vert, off)
        got = prj.persp_proj(mesh, None, None, p, off)
        exp = orast.render(vert, col, G[f"{name}_ind"], p, off)
        np.testing.assert_array_equal(got, exp)
        assert (exp.sum(axis=2) > 0).mean() > 0.2
        # visibility: same winners once the implicit triangle ids are mapped to the filtered positions
        vis = mesh.fetch_visibility()
        ovis = orast.visibility(vert, G[f"{name}_ind"], p, off)
        n = nodata.shape[0]
        full = syn.grid_indices(n)
        keep = np.flatnonzero((~nodata.ravel())[full].all(axis=1))
        np.testing.assert_array_equal(full[keep], G[f"{name}_ind"])
        hit = ovis != 0
        np.testing.assert_array_equal(vis != 0, hit)
        np.testing.assert_array_equal(vis[hit] >> np.uint64(32), ovis[hit] >> np.uint64(32))
        tri_dev = 0xFFFFFFFF - (vis[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        tri_ref = 0xFFFFFFFF - (ovis[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)
        np.testing.assert_array_equal(tri_dev, keep[tri_ref])


def test_set_valid_on_explicit_and_implicit_meshes(L):
    from alproj_amd import project as prj
    n = 64
    s = syn.surface(n)
    rng = np.random.default_rng(3)
    valid = rng.random(n * n) > 0.05
    ind = syn.grid_indices(n)
    kept = ind[valid[ind].all(axis=1)]
    p = syn.base_params(n)
    p.update(w=320, h=200, cx=160.0, cy=100.0, tilt=-20.0, z=p["z"] + 30)
    exp = orast.render(s["vert"], None, kept, p, s["offsets"])
    os.environ["ALP_NO_GRID_DETECT"] = "1"
    try:
        with L.Mesh(s["vert"], None, ind) as m:              # explicit index kernel + mask
            m.set_valid(valid)
            np.testing.assert_array_equal(prj.persp_proj(m, None, None, p, s["offsets"]), exp)
            m.set_valid(None)
            full = prj.persp_proj(m, None, None, p, s["offsets"])
    finally:
        del os.environ["ALP_NO_GRID_DETECT"]
    np.testing.assert_array_equal(full, orast.render(s["vert"], None, ind, p, s["offsets"]))
    with L.Mesh(s["vert"], None, None, grid=(n, n)) as m:     # implicit grid + mask
        m.set_valid(valid)
        np.testing.assert_array_equal(prj.persp_proj(m, None, None, p, s["offsets"]), exp)
    assert (exp != full).any()


def _flat(size=10, elevation=100.0, aerial_dtype=np.uint8, rgb=128):
    """the rasters of the reference's tests/test_surface.py helpers, as arrays"""
    t = (1.0, 0.0, 500.0, 0.0, -1.0, 500.0 + size)
    return (np.full((3, size, size), rgb, dtype=aerial_dtype), np.full((size, size), elevation, dtype=np.float32), t,
            np.zeros((size, size), dtype=bool))


def test_reference_surface_behaviours(L):
    """tests/test_surface.py of the reference: elevation 0 is not nodata, an all-nodata DSM warns
    and draws nothing, a nodata quadrant loses its triangles, dtypes normalise as documented."""
    from alproj_amd import project as prj
    from alproj_amd.surface import colored_surface_mesh
    cam = dict(syn.BASE_CAMERA)
    cam.update(x=505.0, y=505.0, z=40.0, pan=0.0, tilt=-89.0, fov=60.0, w=160, h=120, cx=80.0, cy=60.0)
    aerial, dsm, t, nod = _flat(elevation=0.0)
    mesh, off = colored_surface_mesh(aerial, dsm, t, nod, aerial.dtype)
    with mesh:
        vert, col, valid = mesh.fetch_arrays()
        assert valid.all() and vert.shape == (100, 3) and np.allclose(col, 128 / 255)
        np.testing.assert_array_equal(off, [500.0, 0.0, 501.0])
        full = prj.persp_proj(mesh, None, None, cam, off)
        assert (full[:, :, 0] > 0).mean() > 0.01
    aerial, dsm, t, nod = _flat()
    with pytest.warns(UserWarning, match="All triangles were filtered out"):
        mesh, off = colored_surface_mesh(aerial, dsm * 0, t, ~nod, aerial.dtype, dsm_max_height=0)
    with mesh:
        cam0 = dict(cam, z=40.0)
        assert not prj.persp_proj(mesh, None, None, cam0, off).any()
    nod2 = nod.copy()
    nod2[:5, :5] = True                                     # top-left quadrant (north-west)
    mesh, off = colored_surface_mesh(aerial, dsm, t, nod2, aerial.dtype)
    with mesh:
        _, _, valid = mesh.fetch_arrays()
        np.testing.assert_array_equal(valid.reshape(10, 10), ~nod2)
        cam2 = dict(cam, z=140.0)
        part = prj.persp_proj(mesh, None, None, cam2, off)
        ref_ind = syn.grid_indices(10)
        ref_ind = ref_ind[(~nod2.ravel())[ref_ind].all(axis=1)]
        vert, col, _ = mesh.fetch_arrays()
        np.testing.assert_array_equal(part, orast.render(vert, col, ref_ind, cam2, off))
        covered = (part[:, :, 0] > 0).mean()
        assert 0 < covered < (prj.persp_proj(vert, col, syn.grid_indices(10), cam2, off)[:, :, 0] > 0).mean()
    # dtypes (TestGetColoredSurfaceDtypes): uint16 full scale, float32 in [0, 1], color_max, int16
    for dtype, rgb, kw, exp in [(np.uint16, 32768, {}, 32768 / 65535), (np.float32, 0.5, {}, 0.5),
                                (np.uint16, 2048, {"color_max": 4095}, 2048 / 4095), (np.int16, 16384, {}, 16384 / 32767),
                                (np.float32, -3.0, {}, 0.0)]:
        aerial, dsm, t, nod = _flat(aerial_dtype=dtype, rgb=rgb)
        mesh, off = colored_surface_mesh(aerial, dsm, t, nod, aerial.dtype, **kw)
        with mesh:
            _, col, _ = mesh.fetch_arrays()
            np.testing.assert_array_equal(col, np.float32(exp))


@pytest.mark.parametrize("name", NAMES)
def test_get_colored_surface_io_branch_with_stand_in_rasterio(L, name, monkeypatch):
    """alproj_amd.surface.get_colored_surface end to end, raster I/O included: rasterio (GDAL) is absent, so
    the same stand-ins that fed the REFERENCE's get_colored_surface when g11 was captured
    (tests/golden/gen_golden_surface.py: ``merge`` hands back the seeded rasters, ``fillnodata`` fills with a
    fixed value) feed the product here -- merge, nodata masks, dsm_max_height, fillnodata call, the device
    mesh: same vertices, colours, mask and offsets as the reference returned."""
    import sys
    import types
    from alproj_amd import surface as asurf
    fill_value = 1490.0
    seen = {}

    class FakeDataset:
        def __init__(self, data, nodata, transform):
            self.data, self.nodata, self.transform = data, nodata, transform
            self.dtypes = tuple(str(data.dtype) for _ in range(data.shape[0]))

    def merge(datasets, bounds=None, res=None, resampling=None):
        seen.setdefault("merge_bounds", []).append(bounds)
        return datasets[0].data.copy(), datasets[0].transform

    def fillnodata(arr, mask, max_search_distance=None):
        seen["max_search_distance"] = max_search_distance
        out = arr.copy()
        out[~mask] = fill_value
        return out

    mods = {"rasterio": types.ModuleType("rasterio"), "rasterio.merge": types.ModuleType("rasterio.merge"),
            "rasterio.enums": types.ModuleType("rasterio.enums"), "rasterio.fill": types.ModuleType("rasterio.fill")}
    mods["rasterio.merge"].merge = merge
    mods["rasterio.enums"].Resampling = types.SimpleNamespace(cubic_spline="cubic_spline")
    mods["rasterio.fill"].fillnodata = fillnodata
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    nodata = G[f"{name}_nodata"]
    raw = G[f"{name}_filled"].copy()
    raw[nodata] = np.nan                                   # the DSM as it came out of `merge` when g11 was made
    t = tuple(float(x) for x in G[f"{name}_transform"])
    cm = float(G[f"{name}_color_max"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mesh, col_none, ind_none, off = asurf.get_colored_surface(
            FakeDataset(G[f"{name}_aerial"], None, t), FakeDataset(raw[np.newaxis], None, t), {"x": 0.0, "y": 0.0},
            distance=10, res=1.0, color_max=None if np.isnan(cm) else cm)
    assert col_none is None and ind_none is None
    assert seen["merge_bounds"] == [(-10.0, -10.0, 10.0, 10.0)] * 2 and seen["max_search_distance"] == 300
    with mesh:
        vert, col, valid = mesh.fetch_arrays()
    np.testing.assert_array_equal(off, G[f"{name}_offsets"])
    np.testing.assert_array_equal(vert, G[f"{name}_vert"].astype(np.float32))
    np.testing.assert_array_equal(col, G[f"{name}_col"].astype(np.float32))
    np.testing.assert_array_equal(valid, ~nodata.ravel())


def _filtered_case(n=96, holes=0.04, seed=5):
    s = syn.surface(n)
    rng = np.random.default_rng(seed)
    vvalid = rng.random(n * n) > holes
    vvalid[: 3 * n] &= rng.random(3 * n) > 0.5                   # a ragged first rows: the first kept triangle varies
    full = syn.grid_indices(n)
    kept = full[vvalid[full].all(axis=1)]
    p = syn.base_params(n)
    p.update(w=320, h=200, cx=160.0, cy=100.0, tilt=-20.0, z=p["z"] + 30, k1=-0.05, p2=1e-3)
    return s, vvalid, full, kept, p


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
def test_filtered_grid_index_array_is_recognised(L, dtype):
    """The index array get_colored_surface returns for a DSM with nodata (surface.py:203-205) is the grid
    with triangles removed: it is rendered by the implicit-grid kernels under the vertex mask it implies,
    and the triangle ids handed back are positions in the caller's array."""
    s, vvalid, full, kept, p = _filtered_case()
    ref = orast.visibility(s["vert"], kept, p, s["offsets"])
    derived = np.zeros(len(vvalid), dtype=bool)
    derived[kept.ravel()] = True
    with L.Mesh(s["vert"], None, kept.astype(dtype)) as m:
        assert (m.fetch_arrays()[2] == derived).all() and not derived.all()      # the mask exists: recognised
        m.render_enqueue(L.params_vector(p), s["offsets"])
        np.testing.assert_array_equal(m.fetch_visibility(), ref)
        # a caller's mask on top narrows it; None brings the array's own mask back
        rng = np.random.default_rng(8)
        user = rng.random(len(vvalid)) > 0.03
        m.set_valid(user)
        m.render_enqueue(L.params_vector(p), s["offsets"])
        sub = np.flatnonzero(user[kept].all(axis=1))
        ovis = orast.visibility(s["vert"], kept[sub], p, s["offsets"])
        vis = m.fetch_visibility()
        hit = ovis != 0
        np.testing.assert_array_equal(vis != 0, hit)
        np.testing.assert_array_equal(vis[hit] >> np.uint64(32), ovis[hit] >> np.uint64(32))
        np.testing.assert_array_equal(0xFFFFFFFF - (vis[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64),
                                      sub[0xFFFFFFFF - (ovis[hit] & np.uint64(0xFFFFFFFF)).astype(np.int64)])
        m.set_valid(None)
        m.render_enqueue(L.params_vector(p), s["offsets"])
        np.testing.assert_array_equal(m.fetch_visibility(), ref)
    assert (ref != 0).mean() > 0.3


def test_filtered_grid_image_equals_index_kernels(L, monkeypatch):
    from alproj_amd import project as prj
    s, vvalid, full, kept, p = _filtered_case(n=128, seed=11)
    col = syn.colors(len(s["vert"]))
    exp = orast.render(s["vert"], col, kept, p, s["offsets"], 20.0)
    got = prj.persp_proj(s["vert"], col, kept, p, s["offsets"], 20.0)
    np.testing.assert_array_equal(got, exp)
    monkeypatch.setenv("ALP_NO_GRID_DETECT", "1")
    np.testing.assert_array_equal(prj.persp_proj(s["vert"], col, kept, p, s["offsets"], 20.0), exp)


@pytest.mark.parametrize("case", ["no_mask_explains_it", "shuffled", "duplicate", "flipped", "foreign_triangle", "sparse"])
def test_index_arrays_that_are_not_a_filtered_grid_stay_explicit(L, case):
    s, vvalid, full, kept, p = _filtered_case()
    rng = np.random.default_rng(2)
    if case == "no_mask_explains_it":        # one triangle dropped whose vertices all stay in use
        ind = np.delete(kept, len(kept) // 2, axis=0)
    elif case == "shuffled":
        ind = kept[rng.permutation(len(kept))]
    elif case == "duplicate":
        ind = np.insert(kept, 100, kept[100], axis=0)
    elif case == "flipped":                  # same triangle, other winding / vertex order
        ind = kept.copy()
        ind[777] = ind[777][[0, 2, 1]]
    elif case == "foreign_triangle":
        ind = kept.copy()
        ind[500] = [0, 5, 4000]
    else:                                    # a small part of the grid: cheaper as the array it is
        ind = kept[: len(full) // 5]
    ref = orast.visibility(s["vert"], ind, p, s["offsets"])
    with L.Mesh(s["vert"], None, ind) as m:
        assert m.fetch_arrays()[2].all()                                          # no mask: not converted
        m.render_enqueue(L.params_vector(p), s["offsets"])
        np.testing.assert_array_equal(m.fetch_visibility(), ref)


@pytest.mark.parametrize("shape", [(10, 6), (7, 12), (9, 5), (64, 2), (5, 8), (33, 33)])
@pytest.mark.parametrize("dsm_dtype", [np.float32, np.float64])
def test_mesh_from_rasters_any_shape(L, shape, dsm_dtype):
    """alp_mesh_from_rasters on rasters that are not square (the reference's own index formula only works for square ones,
    quirk Q15; its vertex and colour arrays, surface.py:173-193, 211, are defined for any shape): float64 coordinate minus
    float64 offset then the float32 cast, clamped elevations, normalised colours, the validity mask -- bit for bit the numpy
    expressions"""
    rows, cols = shape
    rng = np.random.default_rng(rows * 100 + cols)
    dsm = (1500 + rng.normal(0, 30, (rows, cols))).astype(dsm_dtype)
    dsm[0, 0] = -5.0                                           # clamped to 0 (surface.py:175)
    aerial = rng.integers(0, 256, (3, rows, cols), dtype=np.uint8)
    nodata = rng.random((rows, cols)) < 0.1
    t = (2.0, 0.0, 732000.0, 0.0, -2.0, 4048000.0 + 2.0 * rows)
    zmax = float(np.float32(1530.0))
    mesh, off = L.Mesh.from_rasters(dsm, t, zmax, aerial, 255.0, nodata)
    with mesh:
        vert, col, valid = mesh.fetch_arrays()
    z = np.clip(dsm.astype(np.float64), 0, zmax)                                       # :175-176
    xx, yy = np.meshgrid(np.arange(cols) * t[0] + t[2], np.arange(rows) * t[4] + t[5])   # :179-181
    ev = np.stack([xx.ravel(), z.ravel(), yy.ravel()], axis=1)                         # :189-190 (X, Z, Y)
    eoff = ev.min(axis=0)                                                              # :211
    np.testing.assert_array_equal(off, eoff)
    np.testing.assert_array_equal(vert, (ev - eoff).astype(np.float32))
    np.testing.assert_array_equal(col, (aerial.reshape(3, -1).T.astype(np.float64) / 255.0).astype(np.float32))
    np.testing.assert_array_equal(valid, ~nodata.ravel())


@pytest.mark.parametrize("aerial_dtype,div", [(np.uint16, 65535.0), (np.float32, 1.0)])
def test_mesh_from_rasters_float64_dsm_with_wide_aerial_types(L, aerial_dtype, div):
    """the float64-DSM instantiations of surface_build_kernel for uint16 and float32 aerial bands (tools/kernel_coverage.py found
    them unlaunched by the suite): the numpy expressions of surface.py:173-193 bit for bit"""
    rows, cols = 21, 34
    rng = np.random.default_rng(77)
    dsm = 900 + rng.normal(0, 12, (rows, cols))                                          # float64
    aerial = (rng.random((3, rows, cols)) * (60000 if aerial_dtype == np.uint16 else 1.0)).astype(aerial_dtype)
    nodata = rng.random((rows, cols)) < 0.15
    t = (1.0, 0.0, 500.0, 0.0, -1.0, 700.0 + rows)
    zmax = 950.0
    mesh, off = L.Mesh.from_rasters(dsm, t, zmax, aerial, div, nodata)
    with mesh:
        vert, col, valid = mesh.fetch_arrays()
    z = np.clip(dsm, 0, zmax)
    xx, yy = np.meshgrid(np.arange(cols) * t[0] + t[2], np.arange(rows) * t[4] + t[5])
    ev = np.stack([xx.ravel(), z.ravel(), yy.ravel()], axis=1)
    np.testing.assert_array_equal(off, ev.min(axis=0))
    np.testing.assert_array_equal(vert, (ev - ev.min(axis=0)).astype(np.float32))
    np.testing.assert_array_equal(col, (aerial.reshape(3, -1).T.astype(np.float64) / div).astype(np.float32))
    np.testing.assert_array_equal(valid, ~nodata.ravel())


def _stand_in_rasterio(monkeypatch, fill_value=900.0):
    import sys
    import types

    def merge(datasets, bounds=None, res=None, resampling=None):
        return datasets[0].data.copy(), datasets[0].transform

    def fillnodata(arr, mask, max_search_distance=None):
        out = arr.copy()
        out[~mask] = fill_value
        return out

    mods = {"rasterio": types.ModuleType("rasterio"), "rasterio.merge": types.ModuleType("rasterio.merge"),
            "rasterio.enums": types.ModuleType("rasterio.enums"), "rasterio.fill": types.ModuleType("rasterio.fill")}
    mods["rasterio.merge"].merge = merge
    mods["rasterio.enums"].Resampling = types.SimpleNamespace(cubic_spline="cubic_spline")
    mods["rasterio.fill"].fillnodata = fillnodata
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)


class _Dataset:
    def __init__(self, data, nodata, transform):
        self.data, self.nodata, self.transform = data, nodata, transform
        self.dtypes = tuple(str(data.dtype) for _ in range(data.shape[0]))


def test_get_colored_surface_integer_rasters_and_its_refusals(L, monkeypatch):
    """the branches of get_colored_surface (surface.py:69-121, 160-171) the g11 rasters do not take: INTEGER rasters whose
    nodata is a value (the DSM's nodata vertices masked and filled, the aerial's zeroed), rasters without any nodata, the
    transform-mismatch refusal and the large-area warning -- against the numpy expressions of the reference"""
    from alproj_amd import surface as asurf
    _stand_in_rasterio(monkeypatch, fill_value=900)
    rng = np.random.default_rng(9)
    rows = cols = 20
    t = (1.0, 0.0, -10.0, 0.0, -1.0, 10.0)
    dsm = rng.integers(800, 1000, (1, rows, cols)).astype(np.int16)
    dsm[0, 3:6, 4:9] = -9999                                       # the DSM's nodata VALUE
    aerial = rng.integers(1, 255, (3, rows, cols)).astype(np.uint8)
    aerial[:, 0, 0] = 255                                           # the aerial's nodata value: zeroed (:102-106)
    mesh, _, _, off = asurf.get_colored_surface(_Dataset(aerial, 255, t), _Dataset(dsm, -9999, t), {"x": 0.0, "y": 0.0}, distance=10, res=1.0)
    with mesh:
        vert, col, valid = mesh.fetch_arrays()
    nod = dsm[0] == -9999
    np.testing.assert_array_equal(valid, ~nod.ravel())
    z = np.where(nod, 900, dsm[0]).astype(np.float64)               # fillnodata's stand-in
    z = np.clip(z, 0, dsm[0][~nod].max())                           # :169, :175-176
    assert off[1] == z.min() and np.array_equal(vert[:, 1], (z.ravel() - z.min()).astype(np.float32))
    a0 = aerial.copy()
    a0[a0 == 255] = 0
    np.testing.assert_array_equal(col, (a0.reshape(3, -1).T.astype(np.float64) / 255.0).astype(np.float32))
    # no nodata anywhere: no mask at all
    dsm2 = rng.integers(800, 1000, (1, rows, cols)).astype(np.int32)
    mesh, _, _, _ = asurf.get_colored_surface(_Dataset(aerial, None, t), _Dataset(dsm2, None, t), {"x": 0.0, "y": 0.0}, distance=10, res=1.0)
    with mesh:
        assert mesh.fetch_arrays()[2].all()
    with pytest.raises(ValueError, match="Transform mismatch"):
        asurf.get_colored_surface(_Dataset(aerial, None, t), _Dataset(dsm2, None, (1.0, 0.0, -11.0, 0.0, -1.0, 10.0)), {"x": 0.0, "y": 0.0}, distance=10, res=1.0)
    with pytest.warns(UserWarning, match="Requested area is very large"):
        mesh, _, _, _ = asurf.get_colored_surface(_Dataset(aerial, None, t), _Dataset(dsm2, None, t), {"x": 0.0, "y": 0.0}, distance=6000, res=1.0)
        mesh.close()
