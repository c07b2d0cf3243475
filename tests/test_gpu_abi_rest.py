"""The entry points of the C ABI that no other GPU test calls (found with ALP_ABI_COVERAGE, tests/conftest.py): the
synchronous render, the (index, xyz) fetch of the visible pixels, the interleaved stand-alone loss, the handle's count, the
timing / event helpers the benchmark uses, device count, and shutdown + re-initialisation (in a process of its own)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import raster as orast
from oracle import ref_numpy as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


def test_counts_events_and_timers(L):
    from alproj_amd import synthetic as syn
    assert L.device_count() >= 1 and L.load().alp_abi_version() == 7
    truth = syn.truth_params(316)
    xyz = syn.gcp_points(20_000, truth, seed=1)
    uv = orc.project_points(xyz, truth)
    with L.Points(xyz, [truth["x"], truth["y"], truth["z"]], "f32") as p:
        n = ctypes.c_int64()
        L.check(L.lib().alp_points_count(p._h, ctypes.byref(n)))
        assert n.value == 20_000 == p.n
        p.set_observed(uv)
        with pytest.raises(L.AlprojHipError) as e:            # nothing evaluated yet: no variant, no shape to report
            p.eval_population_info()
        assert e.value.code == -6
        info = (ctypes.c_int64 * 3)()
        assert L.lib().alp_eval_population_info(None, info) == -1 and L.lib().alp_eval_population_info(p._h, None) == -1
        L.event_record(0)
        p.project(L.params_vector(truth))
        losses, _ = p.eval_population(np.stack([L.params_vector(truth)] * 3), L.LOSS_MEAN_DIST, 0.0)
        # three identical candidates WITH a lens: they share the pose rows -> the shared-pose variant, one stripe per 256 points at most
        variant, stripes, tile_cols = p.eval_population_info()
        assert variant == "shared_pose" and 1 <= stripes <= (20_000 + 255) // 256 and tile_cols == 1
        L.event_record(1)
        L.synchronize()
        assert 0.0 < L.event_elapsed_ms(0, 1) < 1000.0
        k_ms, ar_ms = p.eval_population_timing()
        assert 0.0 < k_ms < 1000.0 and 0.0 <= ar_ms < 1000.0
        assert losses[0] == losses[1] == losses[2] < 1e-2


def test_interleaved_loss_entry_equals_the_column_entry(L):
    rng = np.random.default_rng(2)
    obs = rng.uniform(0, 4000, (50_001, 2))
    prj = obs + rng.normal(0, 7.0, obs.shape)
    for kind, fs, ref in ((L.LOSS_MEAN_DIST, 0.0, orc.mean_distance(obs, prj)), (L.LOSS_HUBER, 10.0, orc.huber(obs, prj, 10.0))):
        a, b = ctypes.c_double(), ctypes.c_double()
        L.check(L.lib().alp_loss_uv(L.as_dp(obs), L.as_dp(prj), len(obs), kind, fs, ctypes.byref(a)))
        u, v, pu, pv = (np.ascontiguousarray(c) for c in (obs[:, 0], obs[:, 1], prj[:, 0], prj[:, 1]))
        L.check(L.lib().alp_loss_uv_columns(L.as_dp(u), L.as_dp(v), L.as_dp(pu), L.as_dp(pv), len(obs), kind, fs, ctypes.byref(b)))
        assert a.value == b.value and a.value == pytest.approx(ref, rel=1e-13)


def test_synchronous_render_and_the_visible_pixel_list(L):
    from alproj_amd import synthetic as syn
    n = 90
    s = syn.surface(n)
    cam = dict(syn.base_params(n), w=240, h=160, cx=120.0, cy=80.0, tilt=-8.0)
    pv = L.params_vector(cam)
    off = np.asarray(s["offsets"], dtype=np.float64)
    with L.Mesh(s["vert"], None, None, grid=(n, n)) as m:
        m.render_enqueue(pv, s["offsets"])
        want = m.fetch()
        out = np.empty((160, 240, 3), dtype=np.float32)
        L.check(L.lib().alp_render(m._h, L.as_dp(pv), L.as_dp(off), 0.0, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
        np.testing.assert_array_equal(out, want)                       # alp_render = enqueue + fetch
        np.testing.assert_allclose(out, orast.render(s["vert"], None, None, cam, s["offsets"], grid=(n, n)), rtol=1e-6, atol=1e-6)
        assert 0.0 < m.frame_ms() < 1000.0
        idx, xyz = m.fetch_valid(s["offsets"])
        seen = np.flatnonzero(want[:, :, 0].ravel() > 0)               # project.py:369: x > 0
        np.testing.assert_array_equal(idx, seen.astype(np.uint32))
        flat = want.reshape(-1, 3)[seen].astype(np.float64)
        np.testing.assert_array_equal(xyz, flat[:, [0, 2, 1]] + off[[0, 2, 1]])        # x, z, y -> x, y, z, plus offsets (project.py:361, 370-373)
        assert len(idx) > 1000


def test_shutdown_and_a_second_life():
    """alp_shutdown releases everything (stream, events, scratch, the fetch's pinned staging and its events) and the library
    can be initialised again in the same process: the second life projects, converts on fetch and renders like the first"""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from alproj_amd import _lib as L
from alproj_amd import synthetic as syn
truth = syn.truth_params(316)
xyz = syn.gcp_points(300_000, truth, seed=3)
res = []
for life in range(2):
    L.init(0)
    with L.Points(xyz, [truth["x"], truth["y"], truth["z"]], "f32") as p:
        p.project(L.params_vector(truth))
        u64, v64 = p.fetch(np.float64)            # the host-pipelined widening fetch: pinned staging + events
        u32, v32 = p.fetch(np.float32)
        assert np.array_equal(u64, u32.astype(np.float64)) and np.array_equal(v64, v32.astype(np.float64))
        res.append(u32.copy())
    L.check(L.load().alp_shutdown())
    assert L.load().alp_synchronize() != 0, "alp_synchronize worked after alp_shutdown"      # ALP_ENOTINIT until the next alp_init
assert np.array_equal(res[0], res[1])
print("two lives ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "two lives ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_the_bindings_refuse_what_the_library_would_misread(L):
    """every shape / type check of alproj_amd._lib (a wrong shape handed to the C side would be read as something else) and
    the integer inputs it converts itself"""
    from alproj_amd import synthetic as syn
    truth = syn.truth_params(316)
    xyz = syn.gcp_points(64, truth, seed=1)
    origin = [truth["x"], truth["y"], truth["z"]]
    for bad in (np.zeros((5, 2)), np.zeros(7), np.zeros((2, 3, 3))):
        with pytest.raises(ValueError, match=r"\(N, 3\)"):
            L.Points(bad, origin)
    with L.Points(np.round(xyz).astype(np.int64), origin, "f64") as pi, L.Points(np.round(xyz), origin, "f64") as pf:      # integer coordinates
        pi.project(L.params_vector(truth)), pf.project(L.params_vector(truth))
        np.testing.assert_array_equal(pi.fetch()[0], pf.fetch()[0])
        uv = np.round(orc.project_points(np.round(xyz), truth))
        pi.set_observed(uv.astype(np.int32)), pf.set_observed(uv)                                                       # integer pixels
        c = np.stack([L.params_vector(truth)])
        np.testing.assert_array_equal(pi.eval_population(c, L.LOSS_HUBER, 10.0)[0], pf.eval_population(c, L.LOSS_HUBER, 10.0)[0])
        with pytest.raises(ValueError, match="observed uv"):
            pi.set_observed(np.zeros((63, 2)))
        with pytest.raises(ValueError, match="observed u, v"):
            pi.set_observed_columns(np.zeros(63), np.zeros(63))
        with pytest.raises(ValueError, match=r"\(P, 25\)"):
            pi.eval_population(np.zeros((3, 24)), L.LOSS_HUBER)
        with pytest.raises(ValueError, match=r"\(B, 25\)"):
            pi.residuals_batch(np.zeros(25))
    with pytest.raises(TypeError, match="unsupported dtype"):
        L.dtype_code(np.zeros(3, np.complex64))
    with pytest.raises(ValueError, match="C-contiguous"):
        L.host_hash64(np.zeros((4, 4))[:, ::2])
    v = np.zeros((9, 3), np.float32)
    with pytest.raises(ValueError, match="vert must have shape"):
        L.Mesh(np.zeros((9, 2), np.float32), None, None, grid=(3, 3))
    with pytest.raises(ValueError, match="value must have the shape"):
        L.Mesh(v, np.zeros((8, 3), np.float32), None, grid=(3, 3))
    with pytest.raises(ValueError, match="either ind or grid"):
        L.Mesh(v)
    with pytest.raises(ValueError, match=r"ind must have shape"):
        L.Mesh(v, None, np.zeros((4, 2), np.int32))
    with L.Mesh(v, None, np.array([[0, 1, 4], [0, 4, 3]], dtype=np.int16)) as m:                                          # a narrow integer index type
        assert m.info()["n_tri"] == 2
        with pytest.raises(ValueError, match=r"\(h, w, 3\)"):
            m.load_image(np.zeros((4, 4), np.float32))
    with pytest.raises(ValueError, match="dsm must have shape"):
        L.Mesh.from_rasters(np.zeros(9), (1, 0, 0, 0, -1, 3), 1.0, np.zeros((3, 3, 3), np.uint8), 255.0)
    with pytest.raises(ValueError, match="nodata must have the shape"):
        L.Mesh.from_rasters(np.zeros((3, 3)), (1, 0, 0, 0, -1, 3), 1.0, np.zeros((3, 3, 3), np.uint8), 255.0, np.zeros((2, 3), bool))
    with pytest.raises(ValueError, match=r"\(n, 3\)"):
        L.distance_mask(np.zeros((4, 2)), [0, 0, 0], 1.0, 2.0)
    with pytest.raises(ValueError, match="x, y"):
        L.rasterize_points_f32(np.zeros(4), np.zeros(5), np.zeros((4, 1)))
    with pytest.raises(ValueError, match="Invalid raster dimensions"):
        L.rasterize_points_f32(np.full(4, 3.0), np.arange(4.0), np.zeros((4, 1)))
    with pytest.raises(ValueError, match="img must be"):
        L.distort_image(np.zeros(5, np.float32), np.zeros(14))
    with pytest.raises(ValueError, match="14 entries"):
        L.distort_image(np.zeros((4, 4), np.float32), np.zeros(13))
    with pytest.raises(ValueError, match="14 entries"):
        L.distort_map(4, 4, np.zeros(3))
    with pytest.raises(ValueError, match="BD must be"):
        L.cma_sample(np.zeros(3), 1.0, np.zeros((3, 2)), None, 4, 10, 1, 0)
    with pytest.raises(ValueError, match="128 bytes"):
        L.comm_init(b"short", 0, 1)
    with pytest.raises(ValueError, match="C-contiguous"):
        L.comm_bcast(np.zeros((4, 4))[:, ::2])


def test_handles_give_their_device_memory_back(L):
    """forty lives of a point set (float32 and float64, row-major and column uploads, observed pixels, projection, both fetches,
    population evaluation with argmin confirmation, residuals) and of a mesh (grid and index array, render, visibility, the
    visible-pixel table, rasterisation from the frame, a mesh from rasters): the free device memory hipMemGetInfo reports after
    them is what it was after the first two (grow-only scratch areas reach their size in those)"""
    from alproj_amd import synthetic as syn
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        L.synchronize()
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    truth = syn.truth_params(316)
    xyz = syn.gcp_points(300_000, truth, seed=2)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(2).normal(0, 1, (len(xyz), 2))
    origin = [truth["x"], truth["y"], truth["z"]]
    cand = np.stack([L.params_vector(truth), L.params_vector(truth), L.params_vector(dict(truth, pan=truth["pan"] + 1e-9))])
    n = 200
    s = syn.surface(n)
    cam = dict(syn.base_params(n), w=480, h=320, cx=240.0, cy=160.0, tilt=-10.0)
    ind = syn.grid_indices(n, np.int32)
    rng = np.random.default_rng(3)
    dsm = (1500 + rng.normal(0, 20, (n, n))).astype(np.float32)
    aerial = rng.integers(0, 256, (3, n, n), dtype=np.uint8)
    img = rng.integers(0, 256, (320, 480, 3), dtype=np.uint8)

    def one_life(k):
        prec = "f32" if k % 2 else "f64"
        pts = L.Points(xyz, origin, prec) if k % 3 else L.Points.from_columns(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), origin, prec)
        with pts as p:
            p.set_observed(uv)
            p.project(cand[0])
            p.fetch(np.float64), p.fetch(np.float32), p.fetch_strided(0, 7, 1000)
            p.eval_population(cand, L.LOSS_HUBER, 10.0)
            if prec == "f64":
                p.residuals(cand[0]), p.residuals_batch(cand)
        os.environ["ALP_NO_GRID_DETECT"] = "1" if k % 2 else "0"
        try:
            mesh = L.Mesh(s["vert"], None, ind if k % 2 else None, grid=None if k % 2 else (n, n))
        finally:
            os.environ.pop("ALP_NO_GRID_DETECT", None)
        with mesh as m:
            m.render_enqueue(L.params_vector(cam), s["offsets"], coords=True)
            m.fetch(), m.fetch_visibility(), m.fetch_valid(s["offsets"])
            cnt, (x0, y0, x1, y1) = m.rasterize_plan(s["offsets"])
            if cnt:
                m.rasterize(img, [0, 1, 2], x0, y1, 2.0, int(np.ceil((x1 - x0) / 2.0)), int(np.ceil((y1 - y0) / 2.0)), 0, 1, 255)
        m2, _ = L.Mesh.from_rasters(dsm, (1.0, 0.0, 0.0, 0.0, -1.0, float(n)), 1600.0, aerial, 255.0, None)
        m2.close()

    one_life(0), one_life(1)
    L.clear_result_pool()
    before = free_bytes()
    for k in range(2, 42):
        one_life(k)
    L.clear_result_pool()
    after = free_bytes()
    assert before - after < (8 << 20), f"{(before - after) / 2**20:.1f} MiB of device memory did not come back after 40 handle lives"
