"""CPU: pin the remap oracle and the wrappers' host logic to vectors captured from the reference's
own ``distort`` / ``reverse_proj`` / ``sim_image`` (tests/golden/gen_golden_render.py: g12, g13).

g13 = the float32 ``map_x`` / ``map_y`` the reference's ``distort`` hands to ``cv2.remap`` (recorded by
a stub in place of cv2): both remap oracles (numpy and C) must reproduce it bit for bit; what
stays unpinned of C4 is only cv2's own nearest rounding and border rule.
g12 = the DataFrame / BGR image the reference's ``reverse_proj`` / ``sim_image`` return for a seeded
raw render (``persp_proj`` replaced by a function returning it)."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import raster as orast
from oracle import ref_numpy as orc

G = os.path.join(os.path.dirname(__file__), "golden")
SIZES = ["48x64", "97x131", "187x281"]
COEFFS = ["identity", "aonly", "radial", "full", "strong"]


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("name", COEFFS)
def test_g13_numpy_oracle_map_is_the_references_map(name, size):
    g = load("g13_distort_map.npz")
    h, w = (int(s) for s in size.split("x"))
    mx, my = orc.distort_maps(w, h, g[f"coeffs_{name}"])
    assert mx.dtype == np.float32
    np.testing.assert_array_equal(mx, g[f"mapx_{name}_{size}"])
    np.testing.assert_array_equal(my, g[f"mapy_{name}_{size}"])
    assert bool(g["interpolation_flag_is_INTER_NEAREST"])


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("name", COEFFS)
def test_g13_c_oracle_gathers_from_the_references_map(name, size):
    """oracle/raster_ref.c computes the map itself (scalar C); an image whose pixel value is its
    own linear index reveals the source pixel it gathers from = rint of the reference's map."""
    g = load("g13_distort_map.npz")
    h, w = (int(s) for s in size.split("x"))
    index_img = (np.arange(h * w, dtype=np.float32) + 1).reshape(h, w, 1)
    got = orast.distort_image(index_img, g[f"coeffs_{name}"])[..., 0]
    sx = np.rint(g[f"mapx_{name}_{size}"].astype(np.float64))
    sy = np.rint(g[f"mapy_{name}_{size}"].astype(np.float64))
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    want = np.where(ok, sy * w + sx + 1, 0).astype(np.float32)
    np.testing.assert_array_equal(got, want)
    if name == "identity":
        np.testing.assert_array_equal(got, index_img[..., 0])


class _HostMesh:
    """Stands in for the device mesh in the host half of reverse_proj: the x > 0 selection of
    alp_render_fetch_valid written in numpy (the device version is tested in test_gpu_golden_render.py)."""
    generation = 1

    def __init__(self, raw):
        self.raw = raw

    def fetch_valid(self, offsets=None):
        flat = self.raw.reshape(-1, 3)
        idx = np.flatnonzero(flat[:, 0] > 0).astype(np.uint32)
        off = np.zeros(3) if offsets is None else np.asarray(offsets, dtype=np.float64)
        xyz = np.stack([flat[idx, 0].astype(np.float64) + off[0], flat[idx, 2].astype(np.float64) + off[2],
                        flat[idx, 1].astype(np.float64) + off[1]], axis=1)
        return idx, xyz


def check_frame(df, g, tag, otag):
    assert list(df.columns) == list(g[f"{tag}_{otag}_columns"])
    assert [str(t) for t in df.dtypes] == list(g[f"{tag}_{otag}_dtypes"])
    np.testing.assert_array_equal(df.index.to_numpy(), g[f"{tag}_{otag}_index"])
    np.testing.assert_array_equal(df.to_numpy(dtype=np.float64), g[f"{tag}_{otag}_values"])


@pytest.mark.parametrize("otag", ["off", "nooff"])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g12_reverse_proj_host_half(tag, otag):
    from alproj_amd import project as aproj
    g = load("g12_wrappers.npz")
    raw, array = g[f"{tag}_raw"], g[f"{tag}_array"]
    off = g["offsets"] if otag == "off" else None
    h, w = raw.shape[:2]
    rp = aproj.ReverseProjection(_HostMesh(raw), off, w, h, False, None)
    check_frame(rp.to_frame(array, list(g[f"{tag}_chnames"])), g, tag, otag)


def test_g12_sim_image_tail_restated():
    """sim_image's tail (project.py:322-324) as the device kernel image_u8_kernel states it -- multiply in float32,
    truncate toward zero, wrap to 8 bits, reverse the channels -- reproduces the reference's bytes (the device
    kernel itself is held to the same fixture in tests/test_gpu_golden_render.py)"""
    g = load("g12_wrappers.npz")
    x = g["sim_raw"] * np.float32(255)
    out = np.ascontiguousarray((np.trunc(x).astype(np.int64) & 0xFF).astype(np.uint8)[:, :, ::-1])
    np.testing.assert_array_equal(out, g["sim_bgr"])


def test_params_vector_accepts_none_principal_point():
    """the reference's project() takes cx = cy = None (intrinsic_mat substitutes w/2, h/2: optimize.py:27-30)"""
    from alproj_amd import _lib
    from alproj_amd import synthetic as syn
    p = dict(syn.base_params(100), cx=None, cy=None)
    v = _lib.params_vector(p)
    assert v[23] == p["w"] / 2 and v[24] == p["h"] / 2
    # ... and only there: None anywhere else fails as the reference's arithmetic does
    for k in ("fov", "w", "h", "x", "k1"):
        with pytest.raises(TypeError):
            _lib.params_vector(dict(p, **{k: None}))
