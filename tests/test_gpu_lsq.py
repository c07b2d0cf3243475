"""Least-squares path on the GPU: batched residual vectors (one launch for the D+1
evaluations of a 2-point finite-difference Jacobian) against the oracle and against scipy's own
sequential finite differences."""
import numpy as np
import pandas as pd
import pytest

from oracle import ref_numpy as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


def _problem(n=900, seed=21, noise=0.5):
    from alproj_amd import synthetic as syn
    truth = syn.truth_params(316)
    xyz = syn.gcp_points(n, truth, seed=seed)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(seed).normal(0, noise, (n, 2))
    return truth, xyz, uv


@pytest.mark.parametrize("prec,tol", [("f64", 1e-9), ("f32", 2e-5)])
def test_residuals_batch_rows(L, prec, tol):
    truth, xyz, uv = _problem()
    rng = np.random.default_rng(4)
    cand = np.tile(L.params_vector(truth), (7, 1))
    cand[1:, 3:7] += rng.uniform(-1, 1, (6, 4))                   # fov, pan, tilt, roll
    cand[1:, 9:11] += rng.uniform(-0.01, 0.01, (6, 2))            # k1, k2
    with L.Points(xyz, [truth["x"], truth["y"], truth["z"]], prec) as pts:
        pts.set_observed(uv)
        got = pts.residuals_batch(cand)
        single = pts.residuals(cand[3])
    assert got.shape == (7, 2 * len(xyz))
    np.testing.assert_array_equal(got[3], single)                 # same arithmetic as the single call
    for b in (0, 3, 6):
        ref = orc.residual_vector(xyz, uv, orc.vector_to_params(cand[b]))
        assert np.all(np.abs(got[b] - ref) <= tol * np.maximum(np.abs(ref), truth["w"]))


def test_lsq_batched_jacobian_matches_scipy_two_point(L):
    from alproj_amd import optimize as opt
    truth, xyz, uv = _problem()
    init = dict(truth, pan=truth["pan"] + 0.8, tilt=truth["tilt"] - 0.6, fov=truth["fov"] + 1.0, k1=0.0, k2=0.0)
    dfx = pd.DataFrame(xyz, columns=["x", "y", "z"])
    dfu = pd.DataFrame(uv, columns=["u", "v"])
    tgt = ["fov", "pan", "tilt", "k1", "k2"]
    o = opt.LsqOptimizer(dfx, dfu, init)
    o.set_target(tgt)
    p_batched, e_batched = o.optimize(method="trf")                       # default: batched Jacobian
    p_scipy, e_scipy = o.optimize(method="trf", jac="2-point")            # scipy's sequential differences
    for k in tgt:
        assert p_batched[k] == pytest.approx(p_scipy[k], rel=1e-6, abs=1e-8), k
    assert e_batched == pytest.approx(e_scipy, rel=1e-9)
    assert e_batched < 0.75 and abs(p_batched["pan"] - truth["pan"]) < 0.02
    # Jacobian itself against scipy's approx_derivative on the oracle residual function
    from scipy.optimize._numdiff import approx_derivative
    bounds = opt.bounds_to_array(init, tgt)
    x0 = np.array([init[k] for k in tgt])

    def f(x):
        return orc.residual_vector(xyz, uv, dict(init, **dict(zip(tgt, x))))

    J_ref = approx_derivative(f, x0, method="2-point", bounds=(bounds[:, 0], bounds[:, 1]))
    pts = o._device_points("f64")
    try:
        J = o._jacobian_function(pts, (bounds[:, 0], bounds[:, 1]))(x0)
    finally:
        pts.close()
    # forward differences with h ~ 1.5e-8 |x| turn the ~1e-9 px agreement of the two float64
    # residual implementations into up to ~0.1 of absolute noise in single entries of J (both
    # Jacobians carry that noise): compare in the Frobenius norm, and bound the outliers
    assert np.linalg.norm(J - J_ref) <= 1e-4 * np.linalg.norm(J_ref)
    assert np.abs(J - J_ref).max() < 0.5
    # robust loss + dogbox also run through the batched path
    p3, e3 = o.optimize(method="dogbox", loss="huber", f_scale=2.0)
    assert abs(p3["pan"] - truth["pan"]) < 0.05
