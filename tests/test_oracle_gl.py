"""CPU: the raster oracle (oracle/raster_ref.c, FROZEN) against the reference's own persp_proj run on a real
OpenGL (g15: Mesa llvmpipe, tests/golden/gen_golden_gl.py).  This is what pins the render: see
tests/gl_compare.py for what is asserted; tests/test_gpu_gl.py repeats it on the HIP path."""
import numpy as np
import pytest

from oracle import raster as orast
from oracle import raycast as oray
from tests import gl_compare as glc
from tests.render_scenes import GL_SCENES


@pytest.fixture(scope="module")
def g15():
    return np.load(glc.G15, allow_pickle=False)


@pytest.mark.parametrize("name", list(GL_SCENES))
def test_raster_oracle_matches_opengl(g15, name):
    s = GL_SCENES[name]()
    p = dict(s["params"], **glc.NO_LENS)
    vis = orast.visibility(s["vert"], s["ind"], p, s["offsets"], grid=s["grid"])
    img = orast.render(s["vert"], s.get("value"), s["ind"], p, s["offsets"], min_distance=s.get("min_distance"), grid=s["grid"])
    r = glc.compare_with_gl(name, s, g15, oray.vis_triangle(vis), img)
    glc.report(name, "raster oracle", r)
    assert r["safe"] > 0.85 * r["pixels"] and r["all_same_rate"] > 0.9998


def test_oracle_sim_image_matches_the_reference_through_opengl(g15):
    """the reference's sim_image (project.py:322-324) through the real GL vs the oracle's render * 255 -> uint8 ->
    BGR: equal bytes on > 99 % of the safe pixels, never more than one level apart (float32 interpolation by
    GL next to a uint8 truncation)"""
    s = GL_SCENES["grid_colours"]()
    p = dict(s["params"], **glc.NO_LENS)
    img = orast.render(s["vert"], s["value"], None, p, s["offsets"], grid=s["grid"])
    sim = np.ascontiguousarray((img * 255).astype(np.uint8)[:, :, ::-1])
    rc = oray.raycast(s["vert"], s["value"], None, p, s["offsets"], grid=s["grid"])
    safe = oray.safe_mask(rc, depth24_steps=glc.DEPTH24_STEPS)[::-1]
    d = np.abs(sim.astype(np.int16) - g15["pair_sim_image"].astype(np.int16))
    assert d[safe].max() <= 1 and (d[safe] == 0).mean() > 0.99


def test_lens_composition_of_the_reference_through_opengl(g15):
    """g15 holds grid_tilt_roll twice as the reference's persp_proj returned it through the real GL: without a lens and
    with one (the only fixture that went through the remap stand-in with a non-identity map).  (1) the reference's own
    pair obeys image_lens = nearest gather of image_plain through the maps of distort() -- i.e. flipud comes BEFORE the
    remap and the maps are applied as restated; (2) so does the raster oracle's pair."""
    from tests.render_scenes import GL_LENS_SCENES
    s = GL_LENS_SCENES["grid_tilt_roll_lens"]()
    glc.check_lens_composition(g15["grid_tilt_roll_lens_image"], g15["grid_tilt_roll_image"], s["params"])
    plain = orast.render(s["vert"], None, None, dict(s["params"], **glc.NO_LENS), s["offsets"], grid=s["grid"])
    lens = orast.render(s["vert"], None, None, s["params"], s["offsets"], grid=s["grid"])
    glc.check_lens_composition(lens, plain, s["params"])


def test_raster_oracle_matches_opengl_at_the_references_frame_size():
    """g16 (tests/golden/gen_golden_gl_c4.py): the reference's persp_proj through the real GL at 5616 x 3744 over 72 M
    triangles -- BASELINE config 4's frame.  The frozen oracle shows GL's triangle on every safe pixel of the fixture's
    1-in-8 lattice and GL's value on its 1-in-16 lattice (a minute of CPU: the float64 ray caster over 72 M triangles)."""
    from tests.render_scenes import C4_SCENES
    g16 = np.load(glc.G16, allow_pickle=False)
    name = "c4_frame_36m"
    s = C4_SCENES[name]()
    p = dict(s["params"], **glc.NO_LENS)
    vis = orast.visibility(s["vert"], None, p, s["offsets"], grid=s["grid"])
    img = orast.render(s["vert"], None, None, p, s["offsets"], grid=s["grid"])
    tri = oray.vis_triangle(vis)
    assert abs(float((tri >= 0).mean()) - float(g16[f"{name}_covered_fraction"])) < 2e-4
    r = glc.compare_with_gl(name, s, g16, tri, img)
    glc.report(name, "raster oracle", r)
    assert r["pixels"] == 468 * 702 and r["safe"] > 0.9 * r["pixels"] and r["all_same_rate"] > 0.999
