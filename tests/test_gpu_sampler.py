"""GPU: the device sampler of the CMA-ES loop (alp_cma_sample) -- the procedure the reference gets from
cmaes.CMA.ask (src/alproj/optimize.py:420-421, bounds handling documented at :381-384): draw, re-draw up
to n_max_resampling times while outside the box, then clip one more draw.  PARITY UNPINNED against cmaes'
random stream (absent, and the reference does not seed it); what is checked is the procedure and the
distribution."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from alproj_amd import _lib
    _lib.init(0)
    return _lib


def _bd(rng, D):
    a = rng.normal(size=(D, D))
    cov = a @ a.T / D + 0.1 * np.eye(D)
    d2, b = np.linalg.eigh(cov)
    return cov, b * np.sqrt(d2)


def test_unbounded_draws_have_the_requested_mean_and_covariance(L):
    rng = np.random.default_rng(0)
    D, P = 9, 200_000
    cov, BD = _bd(rng, D)
    mean = rng.uniform(-1, 1, D)
    x = L.cma_sample(mean, 0.7, BD, None, P, 100, seed=11, generation=3)
    assert x.shape == (P, D) and np.isfinite(x).all()
    np.testing.assert_allclose(x.mean(0), mean, atol=5 * 0.7 * np.sqrt(np.diag(cov).max() / P))
    emp = np.cov(x.T)
    np.testing.assert_allclose(emp, 0.49 * cov, rtol=0.03, atol=0.03 * 0.49 * np.abs(cov).max())
    # marginal normality: standardised first coordinate
    z = (x[:, 0] - mean[0]) / (0.7 * np.sqrt(cov[0, 0]))
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3.0) < 0.08


def test_thirty_dimensions(L):
    """D in (24, 32]: the widest instantiation of cma_sample_kernel (the reference optimises at most 21 parameters; the
    sampler takes up to 32) -- same statistics, same box rule"""
    rng = np.random.default_rng(30)
    D, P = 30, 60_000
    cov, BD = _bd(rng, D)
    mean = rng.uniform(-1, 1, D)
    x = L.cma_sample(mean, 0.5, BD, None, P, 100, seed=3, generation=1)
    assert x.shape == (P, D) and np.isfinite(x).all()
    np.testing.assert_allclose(x.mean(0), mean, atol=6 * 0.5 * np.sqrt(np.diag(cov).max() / P))
    np.testing.assert_allclose(np.cov(x.T), 0.25 * cov, rtol=0.06, atol=0.06 * 0.25 * np.abs(cov).max())
    b = np.column_stack([mean - 0.4, mean + 0.4])
    y = L.cma_sample(mean, 0.5, BD, b, 500, 100, seed=3, generation=1)
    assert ((y >= b[:, 0]) & (y <= b[:, 1])).all()


def test_deterministic_and_keyed_by_seed_generation_candidate(L):
    rng = np.random.default_rng(1)
    D = 21
    _, BD = _bd(rng, D)
    mean = np.full(D, 0.5)
    b = np.column_stack([np.zeros(D), np.ones(D)])
    a1 = L.cma_sample(mean, 0.2, BD, b, 300, 100, seed=5, generation=7)
    a2 = L.cma_sample(mean, 0.2, BD, b, 300, 100, seed=5, generation=7)
    np.testing.assert_array_equal(a1, a2)
    assert not np.array_equal(a1, L.cma_sample(mean, 0.2, BD, b, 300, 100, seed=6, generation=7))
    assert not np.array_equal(a1, L.cma_sample(mean, 0.2, BD, b, 300, 100, seed=5, generation=8))
    # candidate c does not depend on how many candidates are drawn
    np.testing.assert_array_equal(a1[:40], L.cma_sample(mean, 0.2, BD, b, 40, 100, seed=5, generation=7))
    assert len({tuple(r) for r in a1}) == 300


def test_resample_then_clip_rule(L):
    """(a) loose box: everything feasible at the first try; (b) moderate box: the accepted draw is the FIRST
    feasible one of the sequence; (c) hopeless box (the reference's sigma = 1 on [0, 1]^21): n_max tries, then
    the clipped extra draw -- every candidate inside the box with coordinates ON its faces."""
    rng = np.random.default_rng(2)
    D = 21
    BD = np.eye(D)
    mean = np.full(D, 0.5)
    box = np.column_stack([np.zeros(D), np.ones(D)])
    x, tries = L.cma_sample(mean, 0.01, BD, box, 500, 100, seed=1, generation=0, return_tries=True)
    assert (tries == 0).all() and ((x > 0) & (x < 1)).all()
    x, tries = L.cma_sample(mean, 0.25, BD, box, 4000, 100, seed=1, generation=0, return_tries=True)
    assert ((x >= 0) & (x <= 1)).all() and (tries < 100).mean() > 0.99 and tries.max() > 3
    # try t of candidate c is a pure function of (seed, generation, c, t): with n_max = t + 1 ... the same accepted draw
    k = int(np.argmax(tries))
    same = L.cma_sample(mean, 0.25, BD, box, 4000, int(tries[k]) + 1, seed=1, generation=0)
    np.testing.assert_array_equal(same[k], x[k])
    # the acceptance rate per try matches the box probability (geometric number of tries)
    from math import erf, sqrt
    p1 = erf(0.5 / 0.25 / sqrt(2)) ** D
    assert abs((tries == 0).mean() - p1) < 4 * np.sqrt(p1 * (1 - p1) / 4000)
    x, tries = L.cma_sample(mean, 1.0, BD, box, 2048, 100, seed=1, generation=0, return_tries=True)
    assert (tries == 100).all() and ((x >= 0) & (x <= 1)).all()
    on_face = ((x == 0) | (x == 1)).mean()
    assert 0.55 < on_face < 0.70            # P(|N(0,1)| > 0.5) = 0.617 per coordinate
    # n_max_resampling = 0: the first draw is clipped straight away
    x0, t0 = L.cma_sample(mean, 1.0, BD, box, 64, 0, seed=1, generation=0, return_tries=True)
    assert (t0 == 0).all() and ((x0 >= 0) & (x0 <= 1)).all()


def test_cma_with_the_device_sampler_minimises(L):
    """the sampler inside the host CMA class (what CMAOptimizer.optimize uses): a bounded quadratic"""
    from alproj_amd.cma import CMA
    D = 12
    target = np.linspace(0.2, 0.8, D)
    opt = CMA(mean=np.full(D, 0.5), sigma=1.0, bounds=np.column_stack([np.zeros(D), np.ones(D)]), population_size=64,
              n_max_resampling=100, seed=3, sampler=L.cma_sample)
    for _ in range(150):
        X = opt.ask_population()
        assert X.shape == (64, D) and ((X >= 0) & (X <= 1)).all()
        opt.tell_population(X, ((X - target) ** 2).sum(1))
    assert np.abs(opt.mean - target).max() < 1e-3


def test_argument_errors(L):
    with pytest.raises(L.AlprojHipError):
        L.cma_sample(np.zeros(40), 1.0, np.eye(40), None, 4, 10, 0, 0)          # D > 32
    with pytest.raises(L.AlprojHipError):
        L.cma_sample(np.zeros(3), -1.0, np.eye(3), None, 4, 10, 0, 0)
