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
    assert L.device_count() >= 1 and L.load().alp_abi_version() == 6
    truth = syn.truth_params(316)
    xyz = syn.gcp_points(20_000, truth, seed=1)
    uv = orc.project_points(xyz, truth)
    with L.Points(xyz, [truth["x"], truth["y"], truth["z"]], "f32") as p:
        n = ctypes.c_int64()
        L.check(L.lib().alp_points_count(p._h, ctypes.byref(n)))
        assert n.value == 20_000 == p.n
        p.set_observed(uv)
        L.event_record(0)
        p.project(L.params_vector(truth))
        losses, _ = p.eval_population(np.stack([L.params_vector(truth)] * 3), L.LOSS_MEAN_DIST, 0.0)
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
