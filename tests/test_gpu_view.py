"""GPU: the HIP render holds the reference's view / projection scalars (g7) -- the product-side half of
tests/test_oracle_view.py, through alp_render_fetch_visibility on a one-triangle mesh (tests/view_check.py)."""
import os

import numpy as np
import pytest

from oracle import raycast as oray
from tests import view_check as vc

pytestmark = pytest.mark.gpu
G7 = os.path.join(os.path.dirname(__file__), "golden", "g7_gl_matrices.npz")


@pytest.mark.parametrize("k", range(8))
def test_hip_render_uses_the_references_matrices(k):
    from alproj_amd import _lib as L
    L.init(0)
    g = np.load(G7)
    vert, ind, p, exp = vc.scene(g, k)
    with L.Mesh(vert, None, ind) as m:
        m.render_enqueue(L.params_vector(p), vc.OFFSETS_XZY)
        vis = m.fetch_visibility()
    vc.check(oray.vis_triangle(vis), oray.vis_depth(vis), exp, 2e-5)
