"""Small render scenes shared by the CPU oracle tests and the GPU parity tests of the render:
meshes and poses on which the frozen raster oracle (oracle/raster_ref.c), the independent float64
ray caster (oracle/raycast_ref.c) and the HIP kernels are compared."""
import numpy as np

from alproj_amd import synthetic as syn


def _grid_scene(n, w, h, **kw):
    s = syn.surface(n)
    p = dict(syn.base_params(n), w=w, h=h, cx=w / 2.0, cy=h / 2.0)
    p["z"] += kw.pop("dz", 0.0)
    p.update(kw)
    return dict(vert=s["vert"], ind=None, grid=(n, n), params=p, offsets=s["offsets"])


def _hand_made():
    """An explicit-index mesh that is not a grid: two interpenetrating quads and a back-facing
    triangle, seen with tilt and roll (no offsets)."""
    vert = np.array([[-40, 0, 60], [40, 0, 60], [40, 50, 90], [-40, 50, 90],        # sloping quad
                     [-30, 25, 50], [35, 20, 100], [30, 45, 100], [-35, 40, 50],     # quad crossing it
                     [-10, 5, 40], [10, 5, 40], [0, 20, 40]], dtype=np.float32)     # small triangle in front
    ind = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 9]], dtype=np.int32)
    p = dict(syn.BASE_CAMERA, x=3.0, y=-20.0, z=22.0, pan=4.0, tilt=6.0, roll=-9.0, fov=70.0, w=192, h=128,
             cx=96.0, cy=64.0)
    return dict(vert=vert, ind=ind, grid=None, params=p, offsets=None)


SCENES = {
    "grid_tilted": lambda: _grid_scene(120, 320, 200, tilt=-10.0),
    "grid_tilt_roll": lambda: _grid_scene(90, 240, 160, tilt=-12.0, roll=7.0, pan=80.0),
    "grid_near_plane": lambda: _grid_scene(60, 200, 140, dz=-48.0, tilt=-20.0),       # triangles cross vz = 1
    "grid_wide_fov": lambda: _grid_scene(100, 256, 192, fov=88.0, tilt=-30.0, pan=120.0),
    "hand_made_indices": _hand_made,
}


# ---- scenes of the OpenGL fixtures (tests/golden/gen_golden_gl.py -> g15_gl_render.npz) -----------------
# The reference's own persp_proj ran UNMODIFIED on each of these on a real OpenGL (Mesa llvmpipe); the CPU
# oracle and the HIP path are compared with what it returned.  Every scene is rebuilt from seeds here, so
# the fixture stores only GL's outputs.

def _far_scene():
    """terrain 1.2 ... 4.2 km from the camera (2 m cells): the distance the reference is used at
    (example.py:26, docs/usage.md:56), where a 24-bit depth buffer resolves ~1 m"""
    n, res, w, h = 2000, 2.0, 480, 320
    s = syn.surface(n, res=res)
    p = dict(syn.base_params(n, res), w=w, h=h, cx=w / 2.0, cy=h / 2.0, tilt=-8.0, fov=40.0)
    p["z"] += 400.0
    return dict(vert=s["vert"], ind=None, grid=(n, n), params=p, offsets=s["offsets"])


def _nodata_scene():
    """the index array get_colored_surface returns for a DSM with nodata (surface.py:203-205): the grid's
    triangles minus those touching a masked vertex"""
    s = _grid_scene(100, 288, 192, tilt=-14.0, pan=100.0)
    n = 100
    rng = np.random.default_rng(77)
    bad = np.zeros(n * n, dtype=bool)
    for r, c in rng.integers(2, n - 6, (40, 2)):
        bad.reshape(n, n)[r:r + 3, c:c + 4] = True
    ind = syn.grid_indices(n, np.int64)
    s["ind"] = ind[~bad[ind].any(axis=1)]
    s["grid"] = None
    return s


def _with(scene, **kw):
    def make():
        s = SCENES[scene]() if isinstance(scene, str) else scene()
        s.update(kw)
        return s
    return make


def _coloured():
    s = SCENES["grid_tilt_roll"]()
    s["value"] = syn.colors(s["grid"][0] ** 2)
    return s


def _big_cells_scene():
    """a low camera over coarse terrain on a larger frame: cells of 5 ... 100 px next to sub-pixel ones in one image, so that
    the HIP path goes through its LDS depth patches, FAST cells, parked cells and parked triangles in one frame (the small
    scenes above stay on the small-cell paths)"""
    n, res, w, h = 400, 2.0, 960, 640
    s = syn.surface(n, res=res)
    p = dict(syn.base_params(n, res), w=w, h=h, cx=w / 2.0, cy=h / 2.0, tilt=-12.0, fov=62.0, pan=98.0)
    p["z"] -= 22.0                                   # 28 m above the ground
    return dict(vert=s["vert"], ind=None, grid=(n, n), params=p, offsets=s["offsets"])


# scenes whose fixture keeps gl_PrimitiveID for every pixel but GL's image only on every IMAGE_STRIDE-th pixel of both
# axes (the frame is large; the values are asserted on that subset)
IMAGE_STRIDE = {"grid_big_cells": 4, "c4_frame_36m": 16}

LENS = dict(a1=1.03, a2=0.97, k1=-0.06, k2=0.012, k3=0.002, k4=0.004, k5=-0.001, k6=0.0005, p1=0.0015, p2=-0.002,
            s1=0.0006, s2=-0.0002, s3=-0.0004, s4=0.0001)


def _lens_scene():
    """grid_tilt_roll through a lens: the reference's persp_proj ends with flipud (project.py:281) and distort (:292) --
    the one scene whose fixture went through the remap stand-in with a non-identity map (the COMPOSITION is what it pins)"""
    s = SCENES["grid_tilt_roll"]()
    s["params"] = dict(s["params"], **LENS)
    return s


def _c4_frame_scene():
    """BASELINE config 4 at the reference's own sizes: the 5616 x 3744 frame (example.py:22) over a 6000 x 6000 = 36 M-vertex
    surface at 1 m (distance 3000, example.py:25; the grid of examples/pipeline_synthetic.py), the pose of bench.py's render
    leg (camera over the centre of the west edge, fov 75, pan 95) 450 m above the ground and 10 degrees down: 72 M triangles
    at 0.5 ... 6 km -- cells of 8 px (parked cells, LDS depth patches) in the foreground, 5 px at the median pixel, under a
    pixel towards the horizon; every second visible pixel shows a triangle of its own; 38 % of the frame is sky"""
    n, w, h = 6000, 5616, 3744
    s = syn.surface(n)
    p = dict(syn.base_params(n), w=w, h=h, cx=w / 2.0, cy=h / 2.0, tilt=-10.0)
    p["z"] += 400.0
    return dict(vert=s["vert"], ind=None, grid=(n, n), params=p, offsets=s["offsets"])


# the c4-sized fixture (tests/golden/gen_golden_gl_c4.py -> g16_gl_c4_frame.npz) keeps gl_PrimitiveID on every PRIM_STRIDE-th
# pixel of both window axes and GL's image on every IMAGE_STRIDE-th
C4_SCENES = {"c4_frame_36m": _c4_frame_scene}
PRIM_STRIDE = {"c4_frame_36m": 8}

GL_SCENES = dict(SCENES)
GL_SCENES.update({
    "grid_colours": _coloured,                                                # sim_image's call (project.py:322)
    "grid_min_distance": _with("grid_tilted", min_distance=60.0),             # project.py:235,247
    "grid_nodata_indices": _nodata_scene,
    "grid_far_3km": _far_scene,
    "grid_big_cells": _big_cells_scene,
})
GL_LENS_SCENES = {"grid_tilt_roll_lens": _lens_scene}          # compared by their own test (tests/test_oracle_gl.py, test_gpu_gl.py)
