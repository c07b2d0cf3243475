"""GPU: the HIP render against the reference's own persp_proj run on a real OpenGL (g15: Mesa llvmpipe,
tests/golden/gen_golden_gl.py) -- the check that pins C3 to the reference and not to this repository's
reading of the GL specification.  What is asserted: tests/gl_compare.py.  The CPU suite holds the frozen
raster oracle to the same fixtures (tests/test_oracle_gl.py)."""
import numpy as np
import pytest

from oracle import raycast as oray
from tests import gl_compare as glc
from tests.render_scenes import C4_SCENES, GL_SCENES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


@pytest.fixture(scope="module")
def g15():
    return np.load(glc.G15, allow_pickle=False)


@pytest.mark.parametrize("name", list(GL_SCENES))
def test_hip_render_matches_opengl(L, g15, name):
    s = GL_SCENES[name]()
    p = dict(s["params"], **glc.NO_LENS)
    with L.Mesh(s["vert"], s.get("value"), s["ind"], s["grid"]) as m:
        m.render_enqueue(L.params_vector(p), s["offsets"], s.get("min_distance"), coords=s.get("value") is None)
        vis = m.fetch_visibility()
        img = m.fetch()
    r = glc.compare_with_gl(name, s, g15, oray.vis_triangle(vis), img)
    glc.report(name, "HIP", r)
    assert r["safe"] > 0.85 * r["pixels"] and r["all_same_rate"] > 0.9998


def test_reference_render_pair_through_opengl(L, g15):
    """example.py:28,31: sim_image, then reverse_proj of that image at the same pose -- the reference's own
    two wrappers ran through the real GL; the drop-in wrappers must return the same uint8 image (a colour
    interpolated in float32 by GL may land on the other side of a uint8 truncation: <= 1 level, rarely) and
    the same table rows (pixels on triangle edges may see the neighbouring triangle: a few rows per
    thousand appear / disappear; coordinates of common rows agree to GL's sub-pixel snapping)."""
    import pandas as pd
    from alproj_amd import project as aproj
    from alproj_amd import synthetic as syn
    s = GL_SCENES["grid_colours"]()
    n = s["grid"][0]
    ind = syn.grid_indices(n, np.int64)
    vert = s["vert"].astype(np.float64)                     # what get_colored_surface hands over
    sim = aproj.sim_image(vert, s["value"].astype(np.float64), ind, s["params"], s["offsets"])
    ref_sim = g15["pair_sim_image"]
    assert sim.dtype == np.uint8 and sim.shape == ref_sim.shape
    d = np.abs(sim.astype(np.int16) - ref_sim.astype(np.int16))
    rc = oray.raycast(s["vert"], s["value"], None, dict(s["params"], **glc.NO_LENS), s["offsets"], grid=s["grid"])
    safe = oray.safe_mask(rc, depth24_steps=glc.DEPTH24_STEPS)[::-1]
    assert d[safe].max() <= 1 and (d[safe] == 0).mean() > 0.99, (d[safe].max(), (d[safe] == 0).mean())
    df = aproj.reverse_proj(ref_sim, vert, ind, s["params"], s["offsets"])
    ref = pd.DataFrame(g15["pair_reverse_values"], columns=list(g15["pair_reverse_columns"]), index=g15["pair_reverse_index"])
    assert list(df.columns) == list(ref.columns)
    common = df.index.intersection(ref.index)
    assert len(common) > 0.998 * max(len(df), len(ref))
    safe_idx = np.flatnonzero(safe.ravel())
    assert np.isin(safe_idx, df.index).tolist() == np.isin(safe_idx, ref.index).tolist()     # same safe pixels see the surface
    both = np.intersect1d(common, safe_idx)
    a, b = df.loc[both], ref.loc[both]
    np.testing.assert_array_equal(a[["u", "v", "B", "G", "R"]].to_numpy(), b[["u", "v", "B", "G", "R"]].to_numpy())
    err = np.abs(a[["x", "y", "z"]].to_numpy() - b[["x", "y", "z"]].to_numpy())
    print(f"[g15] render pair: sim_image identical on {(d == 0).mean():.5f} of the bytes (max level difference {d.max()}); reverse_proj "
          f"{len(df)} rows vs OpenGL's {len(ref)}, {len(common)} common; max |dxyz| on safe common rows {err.max():.3e} m")
    assert err.max() < 0.05           # metres; cells are 1 m, the sub-pixel spread is asserted per pixel in the scene tests


def test_hip_lens_composition_equals_the_references(L, g15):
    """the fused remap of resolve_kernel composes like the reference's flipud + distort did through the real GL
    (tests/test_oracle_gl.py checks the reference's own pair of images): image with the lens = nearest gather of the
    image without it through distort()'s maps -- and what that gather delivers is GL's image where GL has no freedom"""
    from tests.render_scenes import GL_LENS_SCENES
    s = GL_LENS_SCENES["grid_tilt_roll_lens"]()
    with L.Mesh(s["vert"], None, None, s["grid"]) as m:
        plain = m.render(L.params_vector(dict(s["params"], **glc.NO_LENS)), s["offsets"])
        lens = m.render(L.params_vector(s["params"]), s["offsets"])
        assert m.frame_counts() == (1, 1)                           # the lens is applied by the resolve alone
    glc.check_lens_composition(lens, plain, s["params"])
    # against GL's own lens image: equal wherever the SOURCE pixel is one on which GL and the HIP path agree to 1e-5
    gl_plain, gl_lens = g15["grid_tilt_roll_image"], g15["grid_tilt_roll_lens_image"]
    sy, sx, inside = glc.lens_source(s["params"])
    agree = (np.abs(plain - gl_plain) <= 1e-5 * np.maximum(np.abs(gl_plain), 1.0)).all(axis=2)
    ok = inside & agree[sy, sx]
    assert ok.mean() > 0.6
    assert (np.abs(lens[ok] - gl_lens[ok]) <= 1e-5 * np.maximum(np.abs(gl_lens[ok]), 1.0)).all()


def test_hip_render_matches_opengl_at_the_references_frame_size(L):
    """g16: the reference's own persp_proj on a real OpenGL at BASELINE config 4's frame -- 5616 x 3744 (example.py:22) over a
    6000 x 6000-vertex surface, 72 M triangles, 3 M of them visible -- so that agreement with GL at that size does not rest on
    transitivity through the frozen C oracle.  The fixture keeps gl_PrimitiveID on every 8th pixel of both axes and GL's image on
    every 16th; the assertions are those of the small scenes (tests/gl_compare.py), on that lattice."""
    g16 = np.load(glc.G16, allow_pickle=False)
    name = "c4_frame_36m"
    s = C4_SCENES[name]()
    p = dict(s["params"], **glc.NO_LENS)
    with L.Mesh(s["vert"], None, None, s["grid"]) as m:
        m.render_enqueue(L.params_vector(p), s["offsets"], None, coords=True)
        vis = m.fetch_visibility()
        img = m.fetch()
    tri = oray.vis_triangle(vis)
    assert abs(float((tri >= 0).mean()) - float(g16[f"{name}_covered_fraction"])) < 2e-4       # GL's coverage of the WHOLE frame
    r = glc.compare_with_gl(name, s, g16, tri, img)
    glc.report(name, "HIP", r)
    assert r["pixels"] == 468 * 702 and r["safe"] > 0.9 * r["pixels"] and r["all_same_rate"] > 0.999
