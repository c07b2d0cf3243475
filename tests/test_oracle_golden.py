"""Pin the CPU oracle (oracle/ref_numpy.py) against the golden vectors generated from the
reference's own source (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import ref_numpy as orc

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rtol=1e-12, atol=1e-9)      # float64 restatement vs float64 reference


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_g1_camera_matrices():
    g = load("g1_matrices.npz")
    for pv, K, E in zip(g["params"], g["K"], g["E"]):
        p = orc.vector_to_params(pv)
        np.testing.assert_array_equal(orc.intrinsic_mat(p["fov"], p["w"], p["h"], p["cx"], p["cy"]), K)
        np.testing.assert_array_equal(
            orc.extrinsic_mat(p["pan"], p["tilt"], p["roll"], p["x"], p["y"], p["z"]), E)
    np.testing.assert_array_equal(orc.intrinsic_mat(75, 5616, 3744), g["K_default_75_5616_3744"])
    # SURVEY 8.2 known answer
    K = orc.intrinsic_mat(75, 5616, 3744, 2808, 1872)
    assert abs(K[0, 0] - 3659.45685) < 1e-4 and abs(K[1, 1] - 4014.51696) < 1e-4


@pytest.mark.parametrize("size", ["5616x3744", "641x479"])
@pytest.mark.parametrize("name", ["zero", "aonly", "radial", "full"])
def test_g2_distort(name, size):
    g = load("g2_distort.npz")
    w, h = (int(s) for s in size.split("x"))
    out = orc.distort_points(g[f"pts_{size}"], w, h, *g[f"coeffs_{name}"])
    np.testing.assert_allclose(out, g[f"out_{name}_{size}"], **TOL)


def test_g2_known_answer():
    p = dict(a1=1.02, a2=0.98, k1=-0.05, k2=0.01, k3=0.002, k4=0.003, k5=-0.001, k6=0.0005,
             p1=0.001, p2=-0.002, s1=0.0005, s2=-0.0002, s3=-0.0003, s4=0.0001)
    pts = np.array([[0, 0], [2807.5, 1871.5], [5615, 3743]], dtype=float)
    out = orc.distort_points(pts, 5616, 3744, *[p[k] for k in orc.DIST_KEYS])
    exp = np.array([[123.254273, -2.17983575], [2807.5, 1871.5], [5459.17873, 3721.97324]])
    np.testing.assert_allclose(out, exp, rtol=2e-8)


def test_g3_project():
    g = load("g3_project.npz")
    for i, pv in enumerate(g["params"]):
        uv = orc.project_points(g[f"xyz_{i}"], orc.vector_to_params(pv))
        exp = g[f"uv_{i}"]
        # row 2 is the camera position itself (Q7): NaN for most poses, rounding noise of
        # R.p + R.(-cam) for others -- either way the restatement must reproduce it exactly
        np.testing.assert_array_equal(np.isnan(uv[2]), np.isnan(exp[2]))
        np.testing.assert_allclose(uv, exp, equal_nan=True, **TOL)
    uv = orc.project_points(g["xyz_known"], orc.vector_to_params(g["params_known"]))
    np.testing.assert_allclose(uv, g["uv_known"], equal_nan=True, **TOL)
    assert abs(uv[0, 0] - 2858.677447353897) < 1e-7 and abs(uv[0, 1] - 1922.6842974827396) < 1e-7
    assert abs(uv[3, 0] - 2495.822292923295) < 1e-7       # behind camera: finite, mirrored


def test_g4_losses():
    g = load("g4_losses.npz")
    assert orc.mean_distance(g["obs"], g["proj"]) == pytest.approx(float(g["rmse"]), rel=1e-14)
    assert orc.huber(g["obs"], g["proj"]) == pytest.approx(float(g["huber_default"]), rel=1e-14)
    for tag, f in (("10", 10.0), ("1000", 1000.0), ("0p5", 0.5)):
        assert orc.huber(g["obs"], g["proj"], f) == pytest.approx(float(g[f"huber_{tag}"]), rel=1e-14)


def test_g4_known_answer():
    g = load("g3_project.npz")
    proj = g["uv_known"][:3]
    obs = np.array([[2900, 1700], [3400, 400], [5000, 3500]], dtype=float)
    assert orc.mean_distance(obs, proj) == pytest.approx(493.2229384608617, rel=1e-12)
    assert orc.huber(obs, proj, 10) == pytest.approx(4882.229384608617, rel=1e-12)
    assert orc.huber(obs, proj, 1000) == pytest.approx(166767.20303123057, rel=1e-12)


@pytest.mark.parametrize("name", ["d9", "d12", "d21"])
def test_g5_population(name):
    g = load("g5_population.npz")
    init = orc.vector_to_params(g["params_init"])
    tgt = [str(t) for t in g[f"{name}_targets"]]
    bounds = orc.bounds_to_array(init, tgt)
    np.testing.assert_array_equal(bounds, g[f"{name}_bounds"])
    for tag, fs in (("md", None), ("hub", 10.0)):
        losses, amin = orc.population_losses(g["xyz"], g["uv_obs"], init, tgt, bounds,
                                             g[f"{name}_X"], fs)
        exp = g[f"{name}_{tag}"]
        np.testing.assert_allclose(losses, exp, rtol=1e-12)
        assert amin == int(np.argmin(exp))
        assert losses[3] == losses[7]           # the planted tie


def test_g19_first_phase_population():
    """the reference's first optimisation phase (example.py:19-22, 51-54: a camera without lens coefficients, targets x, y, z, fov,
    pan, tilt, roll, a1, a2): its own `_loss_function` over 140 candidates -- every one of them lens-free"""
    g = load("g19_first_phase.npz")
    init = orc.vector_to_params(g["params_init"])
    assert all(init[k] == 0.0 for k in ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4"))
    tgt = [str(t) for t in g["targets"]]
    bounds = orc.bounds_to_array(init, tgt)
    np.testing.assert_array_equal(bounds, g["bounds"])
    for tag, fs in (("md", None), ("hub", 10.0)):
        losses, amin = orc.population_losses(g["xyz"], g["uv_obs"], init, tgt, bounds, g["X"], fs)
        np.testing.assert_allclose(losses, g[tag], rtol=1e-12)
        assert amin == int(np.argmin(g[tag])) == 11 and losses[3] == losses[7]


def test_g6_bounds():
    g = load("g6_bounds.npz")
    p = orc.vector_to_params(g["params"])
    tgt = [str(t) for t in g["targets"]]
    np.testing.assert_array_equal(orc.bounds_to_array(p, tgt), g["default"])
    np.testing.assert_array_equal(orc.bounds_to_array(p, tgt, {"fov": 10, "cx": 7.5}), g["override"])
    np.testing.assert_array_equal(
        orc.bounds_to_array(p, [str(t) for t in g["all21_targets"]]), g["all21"])


def test_g7_gl_matrices():
    g = load("g7_gl_matrices.npz")
    off = g["cam_offset"]
    for pv, pm, mv in zip(g["params"], g["proj"], g["view"]):
        p = orc.vector_to_params(pv)
        np.testing.assert_array_equal(orc.projection_mat(p["fov"], p["w"], p["h"]), pm)
        np.testing.assert_array_equal(
            orc.modelview_mat(p["pan"], p["tilt"], p["roll"],
                              p["x"] - off[0], p["y"] - off[1], p["z"] - off[2]), mv)
    np.testing.assert_array_equal(
        orc.projection_mat(75, 5616, 3744, near=0.5, far=5000.0, cx=2800.0, cy=1880.0),
        g["proj_cxcy_near_far"])


def test_g8_residuals():
    g = load("g8_residuals.npz")
    r = orc.residual_vector(g["xyz"], g["uv_obs"], orc.vector_to_params(g["params"]))
    np.testing.assert_allclose(r, g["residuals"], **TOL)


def test_remap_nearest_semantics():
    img = np.arange(4 * 5 * 3, dtype=np.float32).reshape(4, 5, 3)
    mx = np.array([[0.5, 1.5, 2.5, -0.6, 4.49]], dtype=np.float32).repeat(4, 0)
    my = np.array([[0, 1, 2, 3]], dtype=np.float32).T.repeat(5, 1)
    out = orc.remap_nearest(img, mx, my)
    # round-half-to-even: 0.5->0, 1.5->2, 2.5->2; -0.6 -> -1 (border 0); 4.49 -> 4
    np.testing.assert_array_equal(out[1, 0], img[1, 0])
    np.testing.assert_array_equal(out[1, 1], img[1, 2])
    np.testing.assert_array_equal(out[1, 2], img[1, 2])
    np.testing.assert_array_equal(out[1, 3], 0)
    np.testing.assert_array_equal(out[1, 4], img[1, 4])
    # identity coefficients -> identity image
    np.testing.assert_array_equal(orc.distort_image(img, [1, 1] + [0] * 12), img)
