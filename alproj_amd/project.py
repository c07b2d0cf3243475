"""Drop-in counterpart of the render half of the reference module ``alproj.project``
(src/alproj/project.py): the depth-buffered mesh render that the reference sends through
OpenGL (moderngl) and the lens-distortion remap it sends through cv2, on an MI355X through
libalproj_hip.so.

============================  ======================================  =====================
here                          reference                               device entry point
============================  ======================================  =====================
``projection_mat``            project.py:13-54                        (host, 16 scalars)
``modelview_mat``             project.py:56-109                       (host, 16 scalars)
``distort``                   project.py:111-143                      alp_distort_image
``persp_proj``                project.py:145-294                      alp_mesh_create + alp_render
``sim_image``                 project.py:296-325                      (persp_proj + uint8/BGR)
``reverse_proj``              project.py:327-374                      alp_render + alp_render_fetch_valid
``reverse_proj_device``       project.py:360 (result kept in HBM)     alp_render_enqueue, alp_render_gather
``rasterize``, ``to_geotiff`` project.py:376-503                      alp_rasterize_columns (file: rasterio)
============================  ======================================  =====================

Extensions (keyword-only, defaults keep the reference behaviour): ``ind=None`` together with
``grid_shape=(rows, cols)`` renders the regular-grid mesh of ``get_colored_surface`` without
an index array; a ``Mesh`` object can be passed as ``vert`` to re-render a device-resident
mesh with new camera parameters (the reference re-uploads everything on every call,
project.py:210-215).

The reference's call pattern -- ``sim_image(vert, col, ind, params, offsets)`` followed by
``reverse_proj(img, vert, ind, params, offsets)`` with the SAME arrays (example.py:28,31; :57,59;
:97,103) -- uploads the mesh on every call, like the reference does (project.py:213-215).  That is the
default here too, because it is the only behaviour that can never show a stale mesh.  Two ways to keep the
mesh on the device between calls:

* pass a ``Mesh`` (or use ``reverse_proj_device``): the explicit device handle, nothing is compared;
* ``set_mesh_cache(True)``: the last mesh stays resident and a call with the same array OBJECTS re-renders
  it -- after the cache has made sure their CONTENT is unchanged: writeable arrays by a 64-bit digest of
  every byte (``alp_host_hash64``, all host cores), taken at upload time and again at lookup: an in-place
  edit of a single vertex between two calls is always seen.  Arrays that are read-only all the way down
  (``a.setflags(write=False)``, no writeable base) are the caller's PROMISE not to edit the memory; they
  are taken by identity plus a sampled digest (every byte up to 32 MB, 16 MB of evenly spaced blocks above),
  which catches the usual way of breaking the promise (flag off, array rewritten, flag on again).  The digest
  costs about what reading the arrays from host memory costs (``bench.py``: ``dropin_call.verify``): it pays
  for writeable arrays only where PCIe is slower than the host's memory, for read-only arrays always.

At an unchanged pose only the resolve stage runs (the library's visibility cache).
"""
import math
import threading
import time
import warnings
import weakref

import numpy as np
import pandas as pd

from . import _lib

_MESH_CACHE = False
_cache = {"mesh": None, "vert": None, "ind": None, "value": None, "grid": None}
LAST_CACHE = {}          # what the last lookup did: hit / miss, seconds spent verifying content


def set_mesh_cache(enabled):
    """Keep the last uploaded mesh on the device for the next call with the same arrays (verified by content, see the
    module docstring).  Off by default: the reference uploads on every call."""
    global _MESH_CACHE
    _MESH_CACHE = bool(enabled)
    if not _MESH_CACHE:
        clear_mesh_cache()


def mesh_cache_enabled():
    return _MESH_CACHE


def _immutable(a):
    """True when nobody can write the array's memory through numpy: read-only, and so is every base it views"""
    while a is not None:
        if not isinstance(a, np.ndarray) or a.flags.writeable:
            return False
        a = a.base
    return True


_SAMPLE_FULL_BELOW = 32 << 20       # read-only arrays up to this size are digested in full (a few ms)
_SAMPLE_BLOCKS, _SAMPLE_BLOCK = 256, 64 << 10


def _sample_digest(a):
    """Digest of a read-only array: every byte up to 32 MB, above that 256 evenly spaced 64 KB blocks (head and tail
    included) -- 16 MB, ~3 ms.  ``writeable=False`` is the caller's promise not to edit the memory; this is the check that
    catches the common way of breaking it (flags toggled, array rewritten, flags toggled back), not a proof: an edit of a
    few rows between two sampled blocks of a large array goes unseen.  Writeable arrays get the full digest."""
    c = np.ascontiguousarray(a)
    if c.nbytes <= _SAMPLE_FULL_BELOW:
        return _lib.host_hash64(c)
    b = c.reshape(-1).view(np.uint8)
    starts = np.linspace(0, b.size - _SAMPLE_BLOCK, _SAMPLE_BLOCKS).astype(np.int64)
    return _lib.host_hash64(np.concatenate([b[s:s + _SAMPLE_BLOCK] for s in starts]))


def _key(a):
    """(weak reference, layout, content digest, "full" | "sample": sampled for an immutable array)"""
    if a is None:
        return None
    layout = (a.shape, a.dtype.str, a.strides, a.__array_interface__["data"][0])
    if _immutable(a):
        return (weakref.ref(a), layout, _sample_digest(a), "sample")
    return (weakref.ref(a), layout, _lib.host_hash64(np.ascontiguousarray(a)), "full")


def _same(key, a):
    if key is None or a is None:
        return key is None and a is None
    if key[0]() is not a or key[1] != (a.shape, a.dtype.str, a.strides, a.__array_interface__["data"][0]):
        return False
    if key[3] == "sample":            # it was immutable when it was uploaded: it still has to be, and look the same
        return _immutable(a) and key[2] == _sample_digest(a)
    return key[2] == _lib.host_hash64(np.ascontiguousarray(a))


def clear_mesh_cache():
    """Drop the device-resident mesh kept for the next call (frees its HBM unless a ReverseProjection still
    holds it)."""
    _cache.update(mesh=None, vert=None, ind=None, value=None, grid=None)


def _new_mesh(vert, value, ind, grid_shape):
    try:
        return _lib.Mesh(vert, value, ind, grid_shape)
    except _lib.AlprojHipError:
        if _cache["mesh"] is None:
            raise
        clear_mesh_cache()                              # the resident mesh and its work areas may be what is in the way
        return _lib.Mesh(vert, value, ind, grid_shape)


def _resident_mesh(vert, value, ind, grid_shape):
    """-> (mesh, owned): the device mesh of these arrays, from the cache when the same arrays were rendered last.
    ``value`` None = the vertices themselves.  ``owned`` meshes are the caller's to close."""
    cacheable = _MESH_CACHE and isinstance(vert, np.ndarray) and (ind is None or isinstance(ind, np.ndarray))
    LAST_CACHE.clear()
    if not cacheable:
        return _new_mesh(vert, value, ind, grid_shape), True
    c = _cache
    t0 = time.perf_counter()
    if c["mesh"] is not None and c["mesh"]._h and c["grid"] == grid_shape and _same(c["vert"], vert) and _same(c["ind"], ind):
        mesh = c["mesh"]
        if value is not None and not _same(c["value"], value):
            c["value"] = None
            mesh.set_value(value)                       # sim_image after reverse_proj: only the colours travel
            c["value"] = _key(value) if isinstance(value, np.ndarray) else None
        LAST_CACHE.update(hit=True, verify_s=time.perf_counter() - t0)
        return mesh, False
    clear_mesh_cache()                                  # before the new upload: both would not have to fit
    t1 = time.perf_counter()
    # the digests are taken by host threads WHILE the arrays cross PCIe (both calls release the GIL); the caller cannot
    # edit the arrays in between: this call has not returned yet
    keys, err = {}, []

    def digest():
        try:
            keys.update(vert=_key(vert), ind=_key(ind), value=_key(value) if isinstance(value, np.ndarray) else None)
        except BaseException as e:          # reported below, in the caller's thread
            err.append(e)

    th = threading.Thread(target=digest)
    th.start()
    try:
        mesh = _lib.Mesh(vert, value, ind, grid_shape)
    finally:
        th.join()
    if err:
        mesh.close()
        raise err[0]
    c.update(mesh=mesh, grid=grid_shape, **keys)
    LAST_CACHE.update(hit=False, verify_s=t1 - t0, upload_and_digest_s=time.perf_counter() - t1)
    return mesh, False


__all__ = ["projection_mat", "modelview_mat", "distort", "persp_proj", "sim_image",
           "reverse_proj", "reverse_proj_device", "ReverseProjection", "rasterize", "to_geotiff", "Mesh",
           "clear_mesh_cache", "set_mesh_cache", "mesh_cache_enabled", "set_timing"]

Mesh = _lib.Mesh


def projection_mat(fov_x_deg, w, h, near=-1, far=1, cx=None, cy=None):
    """OpenGL-style projection matrix as a flat 16-vector (reference project.py:13-54).

    Kept for API compatibility; the render itself folds these numbers on the host exactly as
    the reference's call site does (no cx/cy, near=-1, far=1, untransposed upload)."""
    if cx is None:
        cx = w / 2
    if cy is None:
        cy = h / 2
    fov_x = fov_x_deg * math.pi / 180
    fov_y = fov_x * h / w
    fx = 1 / math.tan(fov_x / 2)
    fy = 1 / math.tan(fov_y / 2)
    return np.array([fx, 0, (w - 2 * cx) / w, 0,
                     0, fy, -(h - 2 * cy) / h, 0,
                     0, 0, -(far + near) / (far - near), -2 * far * near / (far - near),
                     0, 0, -1, 0], dtype=np.float64)


def modelview_mat(pan_deg, tilt_deg, roll_deg, t_x, t_y, t_z):
    """OpenGL-style model-view matrix, transposed and flattened (reference project.py:56-109)."""
    a = (360 - pan_deg) * math.pi / 180
    b = tilt_deg * math.pi / 180
    c = roll_deg * math.pi / 180
    rx = np.array([[1, 0, 0, 0], [0, math.cos(b), -math.sin(b), 0], [0, math.sin(b), math.cos(b), 0], [0, 0, 0, 1.0]])
    ry = np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0], [0, 0, 0, 1.0]])
    rz = np.array([[math.cos(c), -math.sin(c), 0, 0], [math.sin(c), math.cos(c), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    tr = np.array([[1, 0, 0, -t_x], [0, 1, 0, -t_z], [0, 0, 1, -t_y], [0, 0, 0, 1.0]])
    return (rz @ rx @ ry @ tr).T.flatten()


def distort(img, distort_coeffs):
    """Distort an (h, w, c) image with the 14 coefficients a1, a2, k1..k6, p1, p2, s1..s4
    (reference project.py:111-143): nearest-neighbour gather through the inverted-coefficient
    map, zero outside the image.  float32 images go to the device as they are; other dtypes
    are converted to float32 and back (exact for uint8/uint16)."""
    img = np.asarray(img)
    out = _lib.distort_image(img.astype(np.float32, copy=False), distort_coeffs)
    return out if img.dtype == np.float32 else out.astype(img.dtype)


def _params_checked(params):
    if params["fov"] > 90:                     # project.py:208-209 (message kept verbatim)
        warnings.warn("Wider FoV may cause redering fault. Please check the output image carefuly.")
    return _lib.params_vector(params)


def persp_proj(vert, value, ind, params, offsets=None, min_distance=None, *, grid_shape=None):
    """3D -> 2D perspective render of a triangle mesh with per-vertex values (reference
    project.py:145-294).  Returns an (h, w, 3) float32 image, row 0 = top, lens distortion
    applied.

    vert : (N, 3) vertex coordinates in X, Z(vertical), Y order -- or a ``Mesh`` already on
    the device (then ``value`` and ``ind`` are ignored).
    """
    mesh, owned = _enqueue(vert, value, ind, params, offsets, min_distance, grid_shape)
    try:
        t0 = time.perf_counter()
        out = mesh.fetch()
        _fetched(t0)
        return out
    finally:
        if owned:
            mesh.close()


# where the last call spent its time (seconds on the host; `device_ms` = HIP events around the frame's launches):
# `mesh_s` cache lookup or upload, `enqueue_s` launches, `fetch_s` wait + copy back, `frame_s` DataFrame construction
# Filled only after set_timing(True) (benchmarks, probes): reading `device_ms` waits for the frame.
LAST_TIMING = {}
_TIMING = False


def set_timing(enabled):
    """Record where each wrapper call spends its time into LAST_TIMING (off by default: no extra calls, no waits)."""
    global _TIMING
    _TIMING = bool(enabled)
    LAST_TIMING.clear()


def _enqueue(vert, value, ind, params, offsets, min_distance, grid_shape, coords=None):
    """persp_proj up to the finished frame on the device -> (mesh, owned)"""
    pvec = _params_checked(params)
    t0 = time.perf_counter()
    if isinstance(vert, _lib.Mesh):
        mesh, owned, same = vert, False, bool(coords)
    else:
        same = value is vert or value is None
        mesh, owned = _resident_mesh(vert, None if same else value, ind, grid_shape)
    if not _TIMING:
        mesh.render_enqueue(pvec, offsets, min_distance, coords=same)
        return mesh, owned
    t1 = time.perf_counter()
    before = mesh.frame_counts()
    mesh.render_enqueue(pvec, offsets, min_distance, coords=same)
    LAST_TIMING.clear()
    LAST_TIMING.update(mesh_s=t1 - t0, enqueue_s=time.perf_counter() - t1, resident=not owned and before != (0, 0),
                       resolve_only=mesh.frame_counts()[1] > before[1], cache=dict(LAST_CACHE), _mesh=weakref.ref(mesh))
    return mesh, owned


def _fetched(t0, mesh=None):
    if _TIMING:
        m = LAST_TIMING.pop("_mesh", lambda: None)()
        LAST_TIMING.update(fetch_s=time.perf_counter() - t0)
        if m is not None and m._h:
            LAST_TIMING["device_ms"] = m.frame_ms()


def sim_image(vert, color, ind, params, offsets=None, min_distance=None, *, grid_shape=None):
    """Simulated landscape image in OpenCV's BGR uint8 layout (reference project.py:296-325): the ``* 255``,
    ``astype(uint8)`` and RGB -> BGR of :322-324 run on the device (alp_render_fetch_u8), a quarter of the bytes
    of the float32 image cross PCIe."""
    mesh, owned = _enqueue(vert, color, ind, params, offsets, min_distance, grid_shape)
    try:
        t0 = time.perf_counter()
        out = mesh.fetch_u8(255.0, True)
        _fetched(t0)
        return out
    finally:
        if owned:
            mesh.close()


class ReverseProjection:
    """The coordinate image of ``reverse_proj`` kept on the device (SURVEY section 8(f), rows f2 and
    f4): ``to_frame`` builds the reference's DataFrame, ``lookup`` answers ``set_gcp`` for a few
    thousand pixels without ever materialising the ~10 M-row table."""

    def __init__(self, mesh, offsets, w, h, owns_mesh, pvec):
        self.mesh, self.offsets, self.w, self.h, self._owns = mesh, offsets, int(w), int(h), owns_mesh
        self._pvec, self._generation = pvec, mesh.generation

    def _current(self):
        """The mesh may have rendered something else since (it holds one frame): render again."""
        if self.mesh.generation != self._generation:
            self.mesh.render_enqueue(self._pvec, self.offsets, None, coords=True)
            self._generation = self.mesh.generation
        return self.mesh

    def lookup(self, u, v):
        """(n, 3) float64 x, y, z seen at the pixels (u, v) of the simulated image; NaN where the
        pixel is outside the image, not integral, or does not see the surface."""
        u = np.asarray(u, dtype=np.float64)
        v = np.asarray(v, dtype=np.float64)
        whole = (u == np.round(u)) & (v == np.round(v)) & (np.abs(u) < 2**31) & (np.abs(v) < 2**31)
        ui = np.where(whole, u, -1).astype(np.int32)
        vi = np.where(whole, v, -1).astype(np.int32)
        return self._current().gather(ui, vi, self.offsets)

    def to_frame(self, array, chnames=["B", "G", "R"]):
        """The DataFrame of the reference's ``reverse_proj`` (project.py:361-374)."""
        array = np.asarray(array)
        if array.shape[2] != len(chnames):
            raise ValueError("The array has {} channels but chnames has length of {}. Please set chnames correctly."
                             .format(array.shape[2], len(chnames)))
        if array.shape[0] != self.h or array.shape[1] != self.w:
            raise ValueError("all the input array dimensions except for the concatenation axis must match exactly "
                             f"(array is {array.shape[:2]}, the camera image {(self.h, self.w)})")
        # the x > 0 selection (:369), the x,z,y -> x,y,z reorder (:361) and the offsets (:370-373)
        # happen on the device; only the surviving pixels travel back
        t0 = time.perf_counter()
        mesh = self._current()
        C = len(chnames)
        cols = ["x", "y", "z"] + list(chnames)
        if hasattr(mesh, "fetch_valid_table") and C and array.dtype in mesh.TABLE_DTYPES:
            # every column is formed on the device (alp_render_fetch_valid_table): x, y, z and the channels arrive as the
            # contiguous rows of the (3 + C, M) float64 array that BECOMES the DataFrame's block (pandas keeps a block as
            # columns x rows), u, v and the labels as their own arrays -- the host gathers, casts and divides nothing
            labels, u_pix, v_pix, block = mesh.fetch_valid_table(array, self.offsets)
            t1 = time.perf_counter()
            df = pd.DataFrame(block.T, columns=cols, copy=False)
        else:
            if hasattr(mesh, "fetch_valid_block"):
                idx, block = mesh.fetch_valid_block(self.offsets, C)
            else:                               # a host stand-in of the mesh (tests of this host half)
                idx, xyz = mesh.fetch_valid(self.offsets)
                block = _lib.result_empty((3 + C, len(idx)), np.float64)
                block[:3] = xyz.T
            t1 = time.perf_counter()
            flat = array.reshape(-1, array.shape[2])
            if C:
                block[3:] = flat[idx].T         # one gather of the surviving pixels' channels, cast on assignment
            df = pd.DataFrame(block.T, columns=cols, copy=False)
            v_pix, u_pix = np.divmod(idx, np.uint32(self.w))     # one pass: row and column of the linear pixel index
            u_pix, v_pix = u_pix.astype("int16"), v_pix.astype("int16")
            labels = idx.astype(np.int64)
        df.insert(0, "u", u_pix)
        df.insert(1, "v", v_pix)
        # the reference filters a RangeIndex-ed frame, so the labels are the linear pixel indices
        df.index = pd.Index(labels)
        if _TIMING:
            LAST_TIMING.update(fetch_s=t1 - t0, frame_s=time.perf_counter() - t1)
            if LAST_TIMING.pop("_mesh", None) is not None and hasattr(mesh, "frame_ms"):      # this call's own frame (reverse_proj)
                LAST_TIMING["device_ms"] = mesh.frame_ms()
        return df

    def rasterize(self, array, chnames=["B", "G", "R"], resolution=1.0, bands=["R", "G", "B"], interpolate=True,
                  max_dist=1.0, agg_func="mean", nodata=255):
        """``rasterize(self.to_frame(array, chnames), ...)`` -- the compute part of the reference's ``to_geotiff``
        (project.py:414-485) on the table ``reverse_proj`` would return (project.py:361-373) -- without building the
        table: the pixels that see the surface are selected, binned and aggregated on the device straight from the
        resident coordinate image (alp_render_rasterize_plan / alp_render_rasterize); only ``array`` goes up and
        the uint8 raster comes back.  Byte-identical to the table path.  Bands must be channel names."""
        array = np.asarray(array)
        if array.ndim != 3 or array.shape[2] != len(chnames):
            raise ValueError("The array has {} channels but chnames has length of {}. Please set chnames correctly."
                             .format(array.shape[2] if array.ndim == 3 else 1, len(chnames)))
        if array.shape[0] != self.h or array.shape[1] != self.w:
            raise ValueError("all the input array dimensions except for the concatenation axis must match exactly "
                             f"(array is {array.shape[:2]}, the camera image {(self.h, self.w)})")
        columns = ["u", "v", "x", "y", "z"] + list(chnames)
        for band in bands:
            if band not in columns:
                raise ValueError(f"Band '{band}' not found in DataFrame columns: {columns}")
        if agg_func not in _AGG:
            raise ValueError(f"agg_func must be one of {['mean', 'median', 'max', 'min']}")
        if any(b not in chnames for b in bands):          # a coordinate column as a band: through the table
            return rasterize(self.to_frame(array, chnames), resolution, bands, interpolate, max_dist, agg_func, nodata)
        mesh = self._current()
        n, (x_min, y_min, x_max, y_max) = mesh.rasterize_plan(self.offsets)
        if n == 0:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")   # numpy's, as for an empty table
        width = int(np.ceil((x_max - x_min) / resolution))
        height = int(np.ceil((y_max - y_min) / resolution))
        if width <= 0 or height <= 0:
            raise ValueError(f"Invalid raster dimensions: width={width}, height={height}")
        sweeps = int(np.ceil(max_dist / resolution)) if (interpolate and max_dist > 0) else 0
        out = mesh.rasterize(array, [list(chnames).index(b) for b in bands], x_min, y_max, resolution, width, height,
                             _AGG[agg_func], sweeps, nodata)
        return out, (x_min, y_min, x_max, y_max, width, height)

    def to_geotiff(self, array, output_path, chnames=["B", "G", "R"], resolution=1.0, crs="EPSG:6690",
                   bands=["R", "G", "B"], interpolate=True, max_dist=1.0, agg_func="mean", nodata=255):
        """``to_geotiff(reverse_proj(array, ...), output_path, ...)`` of the reference (example.py:103-106) with the
        raster computed from the resident coordinate image (``rasterize`` above); the file is written with rasterio."""
        raster, bounds = self.rasterize(array, chnames, resolution, bands, interpolate, max_dist, agg_func, nodata)
        _write_geotiff(raster, bounds, output_path, crs, bands, nodata)

    def close(self):
        if self.mesh is not None:
            if self._owns:
                self.mesh.close()
            elif getattr(self.mesh, "_h", None) and hasattr(self.mesh, "trim"):
                self.mesh.trim()          # somebody else's mesh (a handle, the opt-in cache): give back the rasterisation work area
        self.mesh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def reverse_proj_device(vert, ind, params, offsets=None, *, grid_shape=None):
    """The render half of ``reverse_proj`` (project.py:360): coordinates of the surface seen by
    every pixel, left on the device as a ``ReverseProjection``."""
    h, w = int(params["h"]), int(params["w"])
    mesh, owns = _enqueue(vert, None, ind, params, offsets, None, grid_shape, coords=True)
    return ReverseProjection(mesh, offsets, w, h, owns, _lib.params_vector(params))


def reverse_proj(array, vert, ind, params, offsets=None, chnames=["B", "G", "R"], *, grid_shape=None):
    """Reverse projection (geo-rectification) of an (h, w, channels) array onto the surface
    (reference project.py:327-374): a DataFrame with u, v, x, y, z and the channels, one row
    per pixel that sees the surface (rows with x > 0 in offset-relative coordinates)."""
    if array.shape[2] != len(chnames):
        raise ValueError("The array has {} channels but chnames has length of {}. Please set chnames correctly."
                         .format(array.shape[2], len(chnames)))
    h, w = int(params["h"]), int(params["w"])
    if array.shape[0] != h or array.shape[1] != w:
        raise ValueError("all the input array dimensions except for the concatenation axis must match exactly "
                         f"(array is {array.shape[:2]}, the camera image {(h, w)})")
    with reverse_proj_device(vert, ind, params, offsets, grid_shape=grid_shape) as rp:
        return rp.to_frame(array, chnames)


_AGG = {"mean": 0, "max": 1, "min": 2, "median": 3}


def rasterize(df, resolution=1.0, bands=["R", "G", "B"], interpolate=True, max_dist=1.0, agg_func="mean",
              nodata=255):
    """The compute part of the reference's ``to_geotiff`` (project.py:414-485) on the device:
    ``reverse_proj`` output -> ``(raster, bounds)`` with ``raster`` a (bands, height, width)
    uint8 array and ``bounds = (x_min, y_min, x_max, y_max, width, height)``.

    Same arguments and error messages as ``to_geotiff``.
    """
    for band in bands:
        if band not in df.columns:
            raise ValueError(f"Band '{band}' not found in DataFrame columns: {list(df.columns)}")
    x = np.ascontiguousarray(df["x"].to_numpy(dtype=np.float64))
    y = np.ascontiguousarray(df["y"].to_numpy(dtype=np.float64))
    (x_min, x_max), (y_min, y_max) = _lib.host_minmax(x), _lib.host_minmax(y)      # one threaded pass each (numpy: four, 14-30 ms)
    width = int(np.ceil((x_max - x_min) / resolution))
    height = int(np.ceil((y_max - y_min) / resolution))
    if width <= 0 or height <= 0:
        raise ValueError(f"Invalid raster dimensions: width={width}, height={height}")
    if agg_func not in _AGG:
        raise ValueError(f"agg_func must be one of {['mean', 'median', 'max', 'min']}")
    # the band columns as they lie in the DataFrame (a float64 column of a block is contiguous: no copy), interleaved on the
    # device -- df[bands].to_numpy() would transpose ~0.4 GB on one core first
    cols = [np.ascontiguousarray(df[b].to_numpy(dtype=np.float64)) for b in bands]
    col_ptrs = (_lib.ctypes.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    sweeps = int(np.ceil(max_dist / resolution)) if (interpolate and max_dist > 0) else 0
    out = _lib.result_empty((len(bands), height, width), np.uint8)
    _lib.check(_lib.lib().alp_rasterize_columns(
        _lib.as_dp(x), _lib.as_dp(y), col_ptrs, len(x), len(bands), float(x_min), float(y_max),
        float(resolution), width, height, _AGG[agg_func], sweeps, int(nodata),
        out.ctypes.data_as(_lib.ctypes.POINTER(_lib.ctypes.c_uint8))))
    return out, (x_min, y_min, x_max, y_max, width, height)


def to_geotiff(df, output_path, resolution=1.0, crs="EPSG:6690", bands=["R", "G", "B"], interpolate=True,
               max_dist=1.0, agg_func="mean", nodata=255):
    """Convert ``reverse_proj`` output to a GeoTIFF (reference project.py:376-503).  The raster
    is computed on the device (``rasterize``); writing the file needs ``rasterio`` like the
    reference does (ImportError otherwise -- use ``rasterize`` to get the arrays)."""
    raster, bounds = rasterize(df, resolution, bands, interpolate, max_dist, agg_func, nodata)
    _write_geotiff(raster, bounds, output_path, crs, bands, nodata)


def _write_geotiff(raster, bounds, output_path, crs, bands, nodata):
    x_min, y_min, x_max, y_max, width, height = bounds
    import rasterio
    from rasterio.transform import from_bounds
    transform = from_bounds(x_min, y_min, x_max, y_max, width, height)
    with rasterio.open(output_path, "w", driver="GTiff", height=height, width=width, count=len(bands),
                       dtype=np.uint8, crs=crs, transform=transform, nodata=nodata) as dst:
        for band_idx in range(len(bands)):
            dst.write(raster[band_idx], band_idx + 1)
    print(f"GeoTIFF saved to {output_path} ({width}x{height} pixels, {len(bands)} bands, nodata={nodata})")
