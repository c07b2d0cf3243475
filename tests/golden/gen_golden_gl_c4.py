#!/usr/bin/env python3
"""Golden vector of the OpenGL render AT BASELINE CONFIG 4's SIZE, from the reference itself on a real OpenGL.

Run only in the build container (reference checkout at /root/reference, Mesa's swrast DRI driver):

    python tests/golden/gen_golden_gl_c4.py            # llvmpipe needs minutes and ~10 GB for this frame

Exactly the set-up of gen_golden_gl.py (the reference's ``persp_proj``, project.py:145-294, UNMODIFIED on Mesa 23.2
llvmpipe through ``_mesa_gl/moderngl_standin.py``; ``cv2.remap`` replaced by a nearest gather that the generator
asserts to be the identity), on ONE scene: ``tests/render_scenes.C4_SCENES["c4_frame_36m"]`` -- the reference's own
frame size 5616 x 3744 (example.py:22) over a 6000 x 6000 = 36 M-vertex, 72 M-triangle surface (example.py:25) --
so that agreement with GL at the size BASELINE config 4 is quoted on does not rest on transitivity through the
frozen C oracle (g15's scenes are at most 960 x 640 pixels and 8 M triangles).

Stored (data only): ``gl_PrimitiveID`` of the draw on every 8th pixel of both window axes, GL's float32 image on
every 16th, GL's strings.  The scene is rebuilt from seeds by the tests.
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "_mesa_gl"))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden as gg                # noqa: E402
import moderngl_standin as mgl         # noqa: E402
from alproj_amd import synthetic as syn          # noqa: E402
from tests.render_scenes import C4_SCENES, IMAGE_STRIDE, PRIM_STRIDE        # noqa: E402


def main():
    opt, prj = gg.load_reference()
    prj.gl = mgl                                   # ``import moderngl as gl`` (project.py:2)
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST = 0
    rec = {}

    def remap(img, map_x, map_y, interpolation=None):
        assert map_x.dtype == np.float32 and interpolation == 0
        h, w = img.shape[:2]
        ix, iy = np.rint(map_x).astype(np.int32), np.rint(map_y).astype(np.int32)
        rec["identity"] = bool((ix == np.arange(w, dtype=np.int32)[None, :]).all() and (iy == np.arange(h, dtype=np.int32)[:, None]).all())
        return img.copy()                          # the identity gather (asserted below)

    cv2.remap = remap
    warnings.simplefilter("ignore")
    info = mgl.gl_info()
    print(info, flush=True)
    out = {"gl_renderer": np.array(info["renderer"]), "gl_version": np.array(info["version"]),
           "gl_subpixel_bits": np.array(info["subpixel_bits"])}
    for name, make in C4_SCENES.items():
        t0 = time.time()
        s = make()
        vert = s["vert"].astype(np.float64)                      # get_colored_surface returns float64 (surface.py:189)
        ind = syn.grid_indices(s["grid"][0], np.int64)           # and an int64 index array (docs/usage.md:96)
        print(f"{name}: scene built in {time.time() - t0:.0f} s: {len(vert)} vertices, {len(ind)} triangles", flush=True)
        mgl.DEPTH_FUNC, mgl.KEEP_DIAGNOSTICS = None, True
        t0 = time.time()
        img = prj.persp_proj(vert, vert, ind, s["params"], s["offsets"])
        print(f"{name}: persp_proj on {info['renderer']} took {time.time() - t0:.0f} s", flush=True)
        assert rec["identity"] and img.dtype == np.float32
        ps, st = PRIM_STRIDE[name], IMAGE_STRIDE[name]
        prim = mgl.LAST["prim_id"]                               # window orientation (row 0 = bottom)
        out[f"{name}_prim_id"] = np.ascontiguousarray(prim[::ps, ::ps]).astype(np.int32)
        out[f"{name}_image"] = np.ascontiguousarray(img[::-1][::st, ::st][::-1])      # strided in WINDOW rows / columns
        out[f"{name}_covered_fraction"] = np.array(float(np.mean(prim >= 0)))
        print(f"{name}: {img.shape[1]}x{img.shape[0]}, covered {np.mean(prim >= 0):.4f}, distinct triangles seen {len(np.unique(prim))}", flush=True)
    path = os.path.join(HERE, "g16_gl_c4_frame.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
