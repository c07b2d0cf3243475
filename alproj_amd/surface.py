"""Drop-in counterpart of the mesh-construction half of the reference module ``alproj.surface``
(src/alproj/surface.py).

``get_colored_surface`` of the reference reads and resamples two rasters (rasterio / GDAL:
``merge``, ``fillnodata``) and then builds, with numpy, a float64 vertex table, a float64 colour
table and an int64 index array -- 120 bytes per vertex that ``persp_proj`` immediately casts to
float32 and uploads again on every call.  Here the raster I/O stays with rasterio (it is not on
the hot path and needs GDAL) and everything after it runs on the device:

* ``colored_surface_mesh``   surface.py:173-212 from arrays -> (device ``Mesh``, offsets)
* ``get_colored_surface``    surface.py:123-212, same signature; returns ``(mesh, None, None,
  offsets)`` so that ``sim_image(vert, col, ind, params, offsets)`` / ``reverse_proj`` keep
  working unchanged with the tuple unpacked (a ``Mesh`` passed as ``vert`` ignores col and ind)
"""
import math
import warnings

import numpy as np

from . import _lib

__all__ = ["color_divisor", "colored_surface_mesh", "get_colored_surface"]


def color_divisor(aerial, source_dtype, color_max=None):
    """The divisor ``_normalize_aerial`` applies (surface.py:44-64); 0.0 = none.  Emits the
    reference's warning for float rasters above 255."""
    source_dtype = np.dtype(source_dtype)
    if color_max is not None:
        return float(color_max)
    if np.issubdtype(source_dtype, np.unsignedinteger) or np.issubdtype(source_dtype, np.signedinteger):
        return float(np.iinfo(source_dtype).max)
    if np.issubdtype(source_dtype, np.floating):
        max_val = float(np.asarray(aerial, dtype=np.float64).max()) if np.asarray(aerial).dtype != np.float32 \
            else float(np.asarray(aerial).max())
        if max_val <= 1.0:
            return 0.0
        if max_val > 255.0:
            warnings.warn(f"Float aerial photo has max value {max_val:.1f} (> 255). "
                          "Dividing by 255; consider passing color_max explicitly.")
        return 255.0
    return 255.0


def colored_surface_mesh(aerial2, dsm2, transform, nodata_mask, source_dtype, color_max=None, dsm_max_height=None):
    """Mesh construction of ``get_colored_surface`` (surface.py:173-212) on the device.

    aerial2 : (>=3, rows, cols) merged aerial bands (nodata already zeroed, :102-106)
    dsm2 : (rows, cols) merged DSM after ``fillnodata`` (:171)
    transform : the affine coefficients (a, b, c, d, e, f) of the merged rasters
    nodata_mask : (rows, cols) bool, True = DSM nodata (:110-117)
    source_dtype : dtype of the aerial raster before merging (:158)
    dsm_max_height : the clamp of :169/:176; default = the largest valid elevation of ``dsm2``

    Returns ``(mesh, offsets)``: an implicit-grid device ``Mesh`` whose vertices are relative to
    ``offsets`` (X, Z, Y order like the reference's).
    """
    aerial2 = np.asarray(aerial2)[:3]
    dsm2 = np.asarray(dsm2)
    nodata_mask = np.asarray(nodata_mask, dtype=bool)
    any_nodata = bool(nodata_mask.any())
    all_nodata = any_nodata and bool(nodata_mask.all())
    if dsm_max_height is None:
        # dsm2[~nodata_mask].max() without the compacted copy of the DSM (50 ms for 6000 x 6000 cells)
        if not any_nodata:
            dsm_max_height = dsm2.max()
        else:
            # `initial` must be a value of the DSM's own type: an int16 DSM cannot hold -inf
            lowest = np.iinfo(dsm2.dtype).min if np.issubdtype(dsm2.dtype, np.integer) else -np.inf
            dsm_max_height = 0 if all_nodata else np.max(dsm2, where=~nodata_mask, initial=lowest)
    if dsm2.min() < 0:
        warnings.warn("DSM still has negative elevation values. Consider using a larger fill_dsm_dist. "
                      "Negative values will be filled with 0.")
    if all_nodata:
        warnings.warn("All triangles were filtered out (all vertices are nodata).")
    div = color_divisor(aerial2, source_dtype, color_max)
    t = [float(transform[k]) for k in range(6)]
    return _lib.Mesh.from_rasters(dsm2, t, float(dsm_max_height), aerial2, div, nodata_mask if any_nodata else None)


def get_colored_surface(aerial, dsm, shooting_point, distance=2000, res=1.0, resampling=None, fill_dsm_dist=300,
                        color_max=None):
    """``alproj.surface.get_colored_surface`` with the mesh built and kept on the device.
    Reading, merging and hole-filling the rasters needs rasterio exactly as in the reference
    (surface.py:69-121, :171); returns ``(mesh, None, None, offsets)``."""
    import rasterio  # noqa: F401  (ImportError here = the raster I/O of the reference is unavailable)
    from rasterio.enums import Resampling
    from rasterio.fill import fillnodata
    from rasterio.merge import merge
    if resampling is None:
        resampling = Resampling.cubic_spline
    source_dtype = aerial.dtypes[0]
    bounds = (shooting_point["x"] - distance, shooting_point["y"] - distance,
              shooting_point["x"] + distance, shooting_point["y"] + distance)
    total_pixels = (2 * distance / res) ** 2
    if total_pixels > 100_000_000:
        warnings.warn(f"Requested area is very large ({total_pixels:.0f} pixels). "
                      "Consider using a larger res or smaller distance.")
    aerial2, transform_a = merge([aerial], bounds=bounds, res=res, resampling=resampling)
    dsm2, transform_d = merge([dsm], bounds=bounds, res=res, resampling=resampling)
    if np.issubdtype(aerial2.dtype, np.floating):
        aerial2[np.isnan(aerial2)] = 0
    elif aerial.nodata is not None:
        aerial2[aerial2 == aerial.nodata] = 0
    if np.issubdtype(dsm2.dtype, np.floating):
        nodata_mask = np.isnan(dsm2[0])
        dsm2[np.isnan(dsm2)] = 0
    elif dsm.nodata is not None:
        nodata_mask = dsm2[0] == dsm.nodata
        dsm2[dsm2 == dsm.nodata] = 0
    else:
        nodata_mask = np.zeros(dsm2.shape[1:], dtype=bool)
    if transform_a != transform_d:
        raise ValueError("Transform mismatch between aerial photo and DSM after merging.")
    dsm_max_height = dsm2[0][~nodata_mask].max() if (~nodata_mask).any() else 0
    filled = fillnodata(dsm2[0], ~nodata_mask, max_search_distance=math.ceil(fill_dsm_dist / res))
    mesh, offsets = colored_surface_mesh(aerial2, filled, tuple(transform_a)[:6], nodata_mask, source_dtype,
                                         color_max=color_max, dsm_max_height=dsm_max_height)
    return mesh, None, None, offsets
