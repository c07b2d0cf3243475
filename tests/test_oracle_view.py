"""CPU: both render oracles hold the reference's view / projection scalars (g7: the reference's own projection_mat /
modelview_mat for 8 poses, tilt x roll != 0 included) -- checked through a rendered triangle, see tests/view_check.py.
tests/test_gpu_view.py does the same for the HIP path."""
import os

import numpy as np
import pytest

from oracle import raster as orast
from oracle import raycast as oray
from tests import view_check as vc

G7 = os.path.join(os.path.dirname(__file__), "golden", "g7_gl_matrices.npz")


@pytest.mark.parametrize("k", range(8))
def test_oracles_use_the_references_matrices(k):
    g = np.load(G7)
    vert, ind, p, exp = vc.scene(g, k)
    vis = orast.visibility(vert, ind, p, vc.OFFSETS_XZY)
    vc.check(oray.vis_triangle(vis), oray.vis_depth(vis), exp, 2e-5)        # float32 1/vz of vertices snapped to 1/256 px
    rc = oray.raycast(vert, None, ind, p, vc.OFFSETS_XZY)
    err = vc.check(rc["tri"], rc["depth"], exp, 1e-9)                        # float64 end to end
    assert err < 1e-9
