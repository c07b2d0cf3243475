#!/usr/bin/env python3
"""Golden vectors of the OpenGL render, FROM THE REFERENCE ITSELF ON A REAL OpenGL.

Run only in the build container (reference checkout at /root/reference, Mesa's swrast DRI driver at
/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so):

    python tests/golden/gen_golden_gl.py

The reference's ``src/alproj/project.py`` is loaded by file path as in gen_golden.py.  Its
``persp_proj`` (project.py:145-294) -- shaders, matrices, buffers, GL state, draw call, read-back,
flip, ``distort`` -- runs UNMODIFIED.  What stands in, and only that:

  ``moderngl``   the PACKAGE (absent) is replaced by ``_mesa_gl/moderngl_standin.py``, a thin object
                 layer that issues the GL calls moderngl would issue on a headless OpenGL 3.3 core
                 context of Mesa 23.2 llvmpipe (``_mesa_gl/dri_ctx.c``).  OpenGL itself -- vertex
                 processing, clipping, rasterisation, the 24-bit depth test, interpolation -- is Mesa's.
  ``cv2.remap``  (opencv absent) by a nearest gather of the float32 maps the reference computes
                 (round half to even, constant 0 outside: OpenCV's documented INTER_NEAREST /
                 BORDER_CONSTANT).  All scenes but ``*_lens`` use the identity lens, for which the
                 gather is the identity whatever the rounding rule; the generator asserts that.
  ``cv2.cvtColor(RGB2BGR)``  by a channel reversal (sim_image only).

Stored per scene of ``tests/render_scenes.GL_SCENES``: the (h, w, 3) float32 image persp_proj returned
under GL's default depth function GL_LESS; as diagnostics from a second draw of the same vertex array
on the same GL: ``gl_PrimitiveID`` per pixel (and GL's 24-bit window depth of the far scene); and the number of pixels
that change under GL_LEQUAL (moderngl's own context default cannot be checked without the package).
For two scenes also what the reference's ``reverse_proj`` / ``sim_image`` return through that GL.
Only data is written: no reference source, shader text or bytecode enters this repository.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "_mesa_gl"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden as gg                # noqa: E402
import moderngl_standin as mgl         # noqa: E402
from alproj_amd import synthetic as syn          # noqa: E402
from tests.render_scenes import GL_LENS_SCENES, GL_SCENES, IMAGE_STRIDE        # noqa: E402


def main():
    opt, prj = gg.load_reference()
    prj.gl = mgl                                   # ``import moderngl as gl`` (project.py:2)
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST, cv2.COLOR_RGB2BGR = 0, 4
    cv2.cvtColor = lambda raw, code: np.ascontiguousarray(raw[:, :, ::-1])
    rec = {}

    def remap(img, map_x, map_y, interpolation=None):
        assert map_x.dtype == np.float32 and interpolation == 0
        ix, iy = np.rint(map_x).astype(np.int64), np.rint(map_y).astype(np.int64)
        h, w = img.shape[:2]
        rec["identity"] = bool((ix == np.arange(w)[None, :]).all() and (iy == np.arange(h)[:, None]).all())
        rec["raw"] = img
        ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
        out = np.zeros_like(img)
        out[ok] = img[iy[ok], ix[ok]]
        return out

    cv2.remap = remap
    warnings.simplefilter("ignore")
    info = mgl.gl_info()
    print(info)
    out = {"gl_renderer": np.array(info["renderer"]), "gl_version": np.array(info["version"]),
           "gl_subpixel_bits": np.array(info["subpixel_bits"])}
    for name, make in GL_SCENES.items():
        s = make()
        vert = s["vert"].astype(np.float64)                      # get_colored_surface returns float64 (surface.py:189)
        value = vert if s.get("value") is None else s["value"].astype(np.float64)
        ind = s["ind"] if s["ind"] is not None else syn.grid_indices(s["grid"][0], np.int64)
        ind = ind.astype(np.int64)
        args = (vert, value, ind, s["params"], s["offsets"], s.get("min_distance"))
        mgl.DEPTH_FUNC, mgl.KEEP_DIAGNOSTICS = None, True
        img = prj.persp_proj(*args)
        assert rec["identity"] and img.dtype == np.float32
        st = IMAGE_STRIDE.get(name, 1)
        out[f"{name}_image"] = np.ascontiguousarray(img[::-1][::st, ::st][::-1]) if st > 1 else img     # strided in WINDOW rows / columns
        out[f"{name}_prim_id"] = mgl.LAST["prim_id"].astype(np.int32)       # window orientation (row 0 = bottom)
        if name == "grid_far_3km":
            out[f"{name}_depth24"] = np.rint(mgl.LAST["depth"].astype(np.float64) * (2 ** 24 - 1)).astype(np.uint32)
        mgl.DEPTH_FUNC, mgl.KEEP_DIAGNOSTICS = "<=", False
        img_le = prj.persp_proj(*args)
        changed = int((img_le != img).any(axis=2).sum())
        out[f"{name}_lequal_changed_pixels"] = np.array(changed)
        mgl.DEPTH_FUNC = None
        print(f"{name}: {img.shape[1]}x{img.shape[0]}, {len(ind)} triangles, covered {np.mean(mgl.LAST['prim_id'] >= 0):.3f}, "
              f"GL_LEQUAL would change {changed} pixels")
    # a lens: the remap stand-in sees a non-identity map (the composition flipud -> distort of project.py:281,292)
    for name, make in GL_LENS_SCENES.items():
        s = make()
        vert = s["vert"].astype(np.float64)
        ind = syn.grid_indices(s["grid"][0], np.int64)
        mgl.DEPTH_FUNC, mgl.KEEP_DIAGNOSTICS = None, False
        img = prj.persp_proj(vert, vert, ind, s["params"], s["offsets"])
        assert not rec["identity"]
        out[f"{name}_image"] = img
        print(f"{name}: {img.shape[1]}x{img.shape[0]}, through the lens")
    # the reference's wrappers through the same GL: the render pair of example.py:28,31 at one pose
    s = GL_SCENES["grid_colours"]()
    n = s["grid"][0]
    ind = syn.grid_indices(n, np.int64)
    vert = s["vert"].astype(np.float64)
    sim = prj.sim_image(vert, s["value"].astype(np.float64), ind, s["params"], s["offsets"])
    out["pair_sim_image"] = sim
    assert sim.dtype == np.uint8
    df = prj.reverse_proj(sim, vert, ind, s["params"], s["offsets"])
    out["pair_reverse_values"] = df.to_numpy(dtype=np.float64)
    out["pair_reverse_index"] = df.index.to_numpy()
    out["pair_reverse_columns"] = np.array(list(df.columns))
    np.savez_compressed(os.path.join(HERE, "g15_gl_render.npz"), **out)
    print("wrote g15_gl_render.npz", os.path.getsize(os.path.join(HERE, "g15_gl_render.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
