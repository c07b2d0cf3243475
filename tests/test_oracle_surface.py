"""oracle.ref_numpy.colored_surface (restating src/alproj/surface.py:168-212) against the
arrays the reference's own get_colored_surface produced (tests/golden/gen_golden_surface.py)."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_surface.npz"))
NAMES = [str(n) for n in G["names"]]


def case(name):
    cm = float(G[f"{name}_color_max"])
    return dict(aerial=G[f"{name}_aerial"], filled=G[f"{name}_filled"], transform=G[f"{name}_transform"],
                nodata=G[f"{name}_nodata"], zmax=float(G[f"{name}_zmax"]), color_max=None if np.isnan(cm) else cm)


@pytest.mark.parametrize("name", NAMES)
def test_colored_surface_restatement(name):
    c = case(name)
    vert, col, ind, off = orc.colored_surface(c["aerial"], c["filled"], c["transform"], c["nodata"],
                                              c["aerial"].dtype, c["color_max"], np.float32(c["zmax"]))
    np.testing.assert_array_equal(off, G[f"{name}_offsets"])
    np.testing.assert_array_equal(vert, G[f"{name}_vert"])
    np.testing.assert_array_equal(col, G[f"{name}_col"])
    np.testing.assert_array_equal(ind, G[f"{name}_ind"])
    assert col.min() >= 0 and col.max() <= 1


def test_color_divisor_rules():
    from alproj_amd.surface import color_divisor
    a = np.zeros((3, 2, 2), np.float32)
    assert color_divisor(a, np.uint8) == 255.0 and color_divisor(a, np.uint16) == 65535.0
    assert color_divisor(a, np.int16) == 32767.0 and color_divisor(a, np.uint16, color_max=4095) == 4095.0
    assert color_divisor(a + 0.5, np.float32) == 0.0
    assert color_divisor(a + 200, np.float32) == 255.0
    with pytest.warns(UserWarning, match="Float aerial photo has max value"):
        assert color_divisor(a + 300, np.float32) == 255.0


# the behaviours the reference's tests/test_surface.py::TestNormalizeAerial pins, on the restatement
@pytest.mark.parametrize("data,dtype,kw,exp", [
    ([[128, 255, 0]], np.uint8, {}, [[128 / 255, 1.0, 0.0]]),
    ([[32768, 65535, 0]], np.uint16, {}, [[32768 / 65535, 1.0, 0.0]]),
    ([[0.5, 1.0, 0.0]], np.float32, {}, [[0.5, 1.0, 0.0]]),
    ([[128.0, 255.0, 0.0]], np.float32, {}, [[128 / 255, 1.0, 0.0]]),
    ([[300.0, 500.0]], np.float32, {}, [[1.0, 1.0]]),
    ([[-10.0, 128.0]], np.uint8, {}, [[0.0, 128 / 255]]),
    ([[500.0, 1000.0]], np.float32, {"color_max": 1000.0}, [[0.5, 1.0]]),
    ([[16384, 32767, 0]], np.int16, {}, [[16384 / 32767, 1.0, 0.0]]),
])
def test_normalize_aerial_rules(data, dtype, kw, exp):
    from alproj_amd.surface import color_divisor
    arr = np.array(data, dtype=np.float64)
    np.testing.assert_allclose(orc.normalize_aerial(arr, np.dtype(dtype), **kw), exp, atol=1e-12)
    # the divisor the device path is given reproduces the same numbers
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        div = color_divisor(arr, dtype, kw.get("color_max"))
    got = np.clip(arr / div if div > 0 else arr, 0, 1)
    np.testing.assert_allclose(got, exp, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.int16, np.uint16, np.int32, np.float32, np.float64])
def test_default_max_height_of_an_integer_dsm_with_partial_nodata(dtype, monkeypatch):
    """colored_surface_mesh(dsm_max_height=None) on an INTEGER DSM with a partial nodata mask raised
    OverflowError (np.max(..., initial=-inf)); the clamp must be dsm2[~nodata_mask].max() as in surface.py:169."""
    from alproj_amd import _lib, surface
    seen = {}

    def recorder(dsm, transform, zmax, aerial, div, nodata):
        seen.update(zmax=zmax, nodata=nodata)
        return "mesh", "offsets"

    monkeypatch.setattr(_lib.Mesh, "from_rasters", staticmethod(recorder))
    rng = np.random.default_rng(3)
    dsm = rng.integers(100, 3000, (12, 9)).astype(dtype)
    mask = np.zeros((12, 9), dtype=bool)
    mask[2:5, 3:7] = True
    dsm[3, 4] = 30000 if dtype != np.int16 else 32000          # the largest value sits under the mask: it must not win
    aerial = np.zeros((3, 12, 9), np.uint8)
    assert surface.colored_surface_mesh(aerial, dsm, (1.0, 0, 0, 0, -1.0, 12.0), mask, np.uint8) == ("mesh", "offsets")
    assert seen["zmax"] == float(dsm[~mask].max()) and seen["nodata"] is mask
    surface.colored_surface_mesh(aerial, dsm, (1.0, 0, 0, 0, -1.0, 12.0), np.zeros_like(mask), np.uint8)
    assert seen["zmax"] == float(dsm.max()) and seen["nodata"] is None
