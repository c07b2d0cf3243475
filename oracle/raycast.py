"""ctypes wrapper of the float64 ray caster (oracle/raycast_ref.c).  TEST INFRASTRUCTURE ONLY.

``raycast`` answers, per pixel centre and independently of any rasterisation rule: nearest
front-facing triangle, its depth, the second-nearest depth, the distance to the nearest projected
edge, the interpolated value.  ``safe_mask`` selects the pixels where every conformant
rasteriser must agree (centre further than ``edge_px`` from every projected edge, first and
second hit separated by more than ``depth_rel`` relative)."""
import ctypes
import os
import subprocess

import numpy as np

from .raster import PARAM_KEYS, _prep

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_oracle", "libalp_raycast.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "raycast_ref.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _lib = ctypes.CDLL(_LIB)
    return _lib


def raycast(vert, value, ind, params, offsets=None, grid=None):
    """-> dict(tri (h,w) int32 [-1 = none], depth, depth2, edge (h,w) float64, value (h,w,3)
    float64), all in GL window orientation (row 0 = bottom)."""
    vert, value, ind, ind_p, i64, n_tri, gh, gw = _prep(vert, value, ind, grid)
    w, h = int(params["w"]), int(params["h"])
    tri = np.empty((h, w), dtype=np.int32)
    depth = np.empty((h, w), dtype=np.float64)
    depth2 = np.empty((h, w), dtype=np.float64)
    edge = np.empty((h, w), dtype=np.float64)
    val = np.empty((h, w, 3), dtype=np.float64)
    pv = np.array([float(params[k]) for k in PARAM_KEYS], dtype=np.float64)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
    vp = ctypes.c_void_p
    rc = lib().alp_raycast(
        vert.ctypes.data_as(vp), None if value is None else value.ctypes.data_as(vp), ctypes.c_int64(vert.shape[0]),
        ind_p, i64, ctypes.c_int64(n_tri), ctypes.c_int64(gh), ctypes.c_int64(gw), pv.ctypes.data_as(vp),
        None if off is None else off.ctypes.data_as(vp), tri.ctypes.data_as(vp), depth.ctypes.data_as(vp),
        depth2.ctypes.data_as(vp), edge.ctypes.data_as(vp), val.ctypes.data_as(vp))
    if rc:
        raise RuntimeError(f"alp_raycast failed: {rc}")
    return dict(tri=tri, depth=depth, depth2=depth2, edge=edge, value=val)


def safe_mask(rc, edge_px=1.0 / 128, depth_rel=1e-4, depth24_steps=None):
    """Pixels whose outcome no implementation-defined rule can change.  With ``depth24_steps`` the first and
    second hit must also lie that many steps of a 24-bit depth buffer apart in the window depth the reference's
    projection produces, 0.5 - 0.5 / vz (project.py:48-53 uploaded untransposed; ~1 m per step at 3 km): the
    criterion for comparisons with a real OpenGL, whose depth renderbuffer is DEPTH_COMPONENT24."""
    hit = rc["tri"] >= 0
    with np.errstate(invalid="ignore", divide="ignore"):
        separated = (rc["depth2"] - rc["depth"]) > depth_rel * rc["depth"]
        if depth24_steps is not None:
            separated &= (0.5 / rc["depth"] - 0.5 / rc["depth2"]) * (2 ** 24 - 1) > depth24_steps
    return (rc["edge"] > edge_px) & (~hit | separated)


def vis_triangle(vis):
    """Triangle index (int64, -1 = background) out of a 64-bit visibility buffer."""
    vis = np.asarray(vis, dtype=np.uint64)
    tri = (np.uint64(0xFFFFFFFF) - (vis & np.uint64(0xFFFFFFFF))).astype(np.int64)
    return np.where(vis == 0, -1, tri)


def vis_depth(vis):
    """View depth vz (float64) encoded in a 64-bit visibility buffer (float32 bits of 1/vz)."""
    bits = (np.asarray(vis, dtype=np.uint64) >> np.uint64(32)).astype(np.uint32)
    with np.errstate(divide="ignore"):
        return 1.0 / bits.view(np.float32).astype(np.float64)
