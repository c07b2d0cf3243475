"""Drop-in counterpart of the two GCP-table functions of the reference module ``alproj.gcp``
that sit between the render and the optimiser:

* ``set_gcp``              src/alproj/gcp.py:614-648
* ``filter_gcp_distance``  src/alproj/gcp.py:651-726

Feature matching itself (``image_match`` and friends: CNN matchers, RANSAC) is a different
workload and stays with the reference.  ``set_gcp`` accepts, besides the reference's
``reverse_proj`` DataFrame, the device-resident ``alproj_amd.project.ReverseProjection``: the
(u_sim, v_sim) -> (x, y, z) join then becomes one gather from the coordinate image in HBM
(``alp_render_gather``) and the multi-million-row table is never built.
"""
import numpy as np
import pandas as pd

from . import _lib
from .project import ReverseProjection

__all__ = ["set_gcp", "filter_gcp_distance"]


def set_gcp(match, rev_proj):
    """Add geographic coordinates to matched point pairs (reference gcp.py:614-648).

    match : DataFrame with u_org, v_org, u_sim, v_sim (result of image_match).
    rev_proj : DataFrame of ``reverse_proj`` -- or a ``ReverseProjection`` on the device.

    Returns a DataFrame u, v, x, y, z: the matches whose simulated-image pixel sees the surface
    (left join on (u_sim, v_sim) = (u, v), rows with any NaN dropped, labels of the join kept)."""
    if isinstance(rev_proj, ReverseProjection):
        xyz = rev_proj.lookup(match["u_sim"].to_numpy(), match["v_sim"].to_numpy())
        gcp = pd.DataFrame({"u": match["u_org"].to_numpy(), "v": match["v_org"].to_numpy(),
                            "x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]})
    else:
        gcp = pd.merge(match, rev_proj, how="left", left_on=["u_sim", "v_sim"], right_on=["u", "v"])
        gcp = gcp[["u_org", "v_org", "x", "y", "z"]].rename(columns={"u_org": "u", "v_org": "v"})
    return gcp.dropna(how="any", axis=0)


def filter_gcp_distance(gcp, params, min_distance=None, max_distance=None):
    """Keep the GCPs whose 3-D distance from the camera position lies in
    [min_distance, max_distance] (reference gcp.py:651-726): same validation and messages, rows
    with NaN coordinates dropped, index reset; a copy when there is nothing to filter.  The mask is
    formed on the device (``alp_distance_mask``), next to the gather that made the coordinates."""
    for key in ("x", "y", "z"):
        if key not in params:
            raise KeyError(f"params must contain '{key}' key")
    if min_distance is not None and min_distance < 0:
        raise ValueError("min_distance must be non-negative")
    if min_distance is not None and max_distance is not None and max_distance < min_distance:
        raise ValueError("max_distance must be >= min_distance")
    if len(gcp) == 0 or (min_distance is None and max_distance is None):
        return gcp.copy()
    xyz = np.column_stack([gcp["x"].to_numpy(dtype=np.float64), gcp["y"].to_numpy(dtype=np.float64),
                           gcp["z"].to_numpy(dtype=np.float64)])
    keep = _lib.distance_mask(xyz, [params["x"], params["y"], params["z"]], min_distance, max_distance)
    return gcp[keep].reset_index(drop=True)
