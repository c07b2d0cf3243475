"""ctypes wrapper of the C raster oracle (oracle/raster_ref.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_oracle", "libalp_oracle.so")
_lib = None

PARAM_KEYS = ("x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2",
              "k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2",
              "s1", "s2", "s3", "s4", "w", "h", "cx", "cy")


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "raster_ref.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _pvec(params):
    return np.array([float(params[k]) for k in PARAM_KEYS], dtype=np.float64)


def _prep(vert, value, ind, grid):
    vert = np.ascontiguousarray(vert, dtype=np.float32)
    value = None if value is None else np.ascontiguousarray(value, dtype=np.float32)
    if ind is None:
        gh, gw = grid
        n_tri = 2 * (gh - 1) * (gw - 1)
        ind_p, i64 = None, 0
    else:
        ind = np.ascontiguousarray(ind)
        if ind.dtype not in (np.int32, np.int64):
            ind = ind.astype(np.int64)
        gh = gw = 0
        n_tri = ind.shape[0]
        ind_p, i64 = ind.ctypes.data_as(ctypes.c_void_p), int(ind.dtype == np.int64)
    return vert, value, ind, ind_p, i64, n_tri, gh, gw


def visibility(vert, ind, params, offsets=None, grid=None):
    """(h, w) uint64 visibility buffer in GL window orientation (row 0 = bottom)."""
    vert, _, ind, ind_p, i64, n_tri, gh, gw = _prep(vert, None, ind, grid)
    w, h = int(params["w"]), int(params["h"])
    vis = np.zeros((h, w), dtype=np.uint64)
    pv = _pvec(params)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
    rc = lib().alp_ref_visibility(
        vert.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(vert.shape[0]), ind_p, i64, ctypes.c_int64(n_tri),
        ctypes.c_int64(gh), ctypes.c_int64(gw), pv.ctypes.data_as(ctypes.c_void_p),
        None if off is None else off.ctypes.data_as(ctypes.c_void_p), vis.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError(f"alp_ref_visibility failed: {rc}")
    return vis


def render(vert, value, ind, params, offsets=None, min_distance=None, grid=None):
    """persp_proj restatement: (h, w, 3) float32, row 0 = top."""
    vert, value, ind, ind_p, i64, n_tri, gh, gw = _prep(vert, value, ind, grid)
    w, h = int(params["w"]), int(params["h"])
    out = np.zeros((h, w, 3), dtype=np.float32)
    pv = _pvec(params)
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
    rc = lib().alp_ref_render(
        vert.ctypes.data_as(ctypes.c_void_p), None if value is None else value.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(vert.shape[0]), ind_p, i64, ctypes.c_int64(n_tri), ctypes.c_int64(gh), ctypes.c_int64(gw),
        pv.ctypes.data_as(ctypes.c_void_p), None if off is None else off.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_double(0.0 if min_distance is None else float(min_distance)), out.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError(f"alp_ref_render failed: {rc}")
    return out


def distort_image(img, coeffs):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    out = np.zeros_like(img)
    cf = np.ascontiguousarray(coeffs, dtype=np.float64)
    lib().alp_ref_distort_image(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(h), ctypes.c_int64(w),
                                ctypes.c_int64(c), cf.ctypes.data_as(ctypes.c_void_p),
                                out.ctypes.data_as(ctypes.c_void_p))
    return out
