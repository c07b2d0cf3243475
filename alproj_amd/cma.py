"""Host-side (mu/mu_w, lambda)-CMA-ES sampler with box constraints.

The reference drives the third-party package ``cmaes`` (pinned ``cmaes==0.12.0`` in its
requirements.txt:14; ``>=0.8`` in pyproject.toml:35), which is NOT part of the reference
checkout and is not installed here.  This module restates the published algorithm
(N. Hansen, "The CMA Evolution Strategy: A Tutorial", arXiv:1604.00772, with active
covariance update / negative recombination weights, as implemented by ``cmaes.CMA``) behind the
contract the reference relies on at src/alproj/optimize.py:410-427:

* ``CMA(mean, sigma, bounds, population_size, n_max_resampling[, seed])``
* ``ask()`` -> one candidate inside ``bounds``: re-sample up to ``n_max_resampling`` times,
  then clip the last sample (documented at optimize.py:381-384)
* ``tell(solutions)`` with ``solutions = [(x, value), ...]``: sorts the list IN PLACE by value
  (stable), which is why the reference can read the best candidate as ``solutions[0][0]``
  afterwards (optimize.py:424-427, quirk Q9).

PARITY UNPINNED for the sampler: the reference constructs CMA without a seed, so its
trajectory is not reproducible, and cmaes itself is absent; parity of the optimiser is
defined at the evaluation boundary (same candidates -> same losses / argmin).

``ask_population()`` is the vectorised form used by the GPU path: the whole generation is
sampled at once so that it can be evaluated by ONE kernel launch.  With ``sampler=`` (the device
sampler ``alproj_amd._lib.cma_sample`` = ``alp_cma_sample``) the draws themselves happen on the GPU:
the reference's default sigma = 1.0 makes almost every draw infeasible, i.e. 101 draws per candidate.
"""
import math

import numpy as np

_EPS = 1e-8
_MEAN_MAX = 1e32
_SIGMA_MAX = 1e32


class CMA:
    def __init__(self, mean, sigma, bounds=None, n_max_resampling=100, seed=None,
                 population_size=None, cov=None, sampler=None):
        mean = np.asarray(mean, dtype=np.float64)
        if sigma <= 0:
            raise ValueError("sigma must be non-zero positive value")
        if not np.all(np.abs(mean) < _MEAN_MAX):
            raise ValueError(f"Abs of all elements of mean vector must be less than {_MEAN_MAX}")
        n = len(mean)
        if n < 1:
            raise ValueError("The dimension of mean must be positive")
        if population_size is None:
            population_size = 4 + math.floor(3 * math.log(n))
        if population_size <= 0:
            raise ValueError("popsize must be non-zero positive value.")
        lam = int(population_size)
        mu = lam // 2

        w_prime = np.array([math.log((lam + 1) / 2) - math.log(i + 1) for i in range(lam)])
        mu_eff = (np.sum(w_prime[:mu]) ** 2) / np.sum(w_prime[:mu] ** 2) if mu > 0 else 1.0
        neg = w_prime[mu:]
        mu_eff_minus = (np.sum(neg) ** 2) / np.sum(neg ** 2) if neg.size and np.sum(neg ** 2) > 0 else 0.0

        alpha_cov = 2.0
        c1 = alpha_cov / ((n + 1.3) ** 2 + mu_eff)
        cmu = min(1 - c1 - 1e-8,
                  alpha_cov * (mu_eff - 2 + 1 / mu_eff) / ((n + 2) ** 2 + alpha_cov * mu_eff / 2))
        if not (c1 <= 1 - cmu and cmu <= 1 - c1):
            raise ValueError("invalid learning rates")
        min_alpha = min(1 + c1 / cmu if cmu > 0 else np.inf,
                        1 + (2 * mu_eff_minus) / (mu_eff + 2),
                        (1 - c1 - cmu) / (n * cmu) if cmu > 0 else np.inf)
        pos_sum = np.sum(w_prime[w_prime > 0])
        neg_sum = np.sum(np.abs(w_prime[w_prime < 0]))
        weights = np.where(w_prime >= 0, w_prime / pos_sum,
                           (min_alpha / neg_sum if neg_sum > 0 else 0.0) * w_prime)

        c_sigma = (mu_eff + 2) / (n + mu_eff + 5)
        d_sigma = 1 + 2 * max(0, math.sqrt((mu_eff - 1) / (n + 1)) - 1) + c_sigma
        cc = (4 + mu_eff / n) / (n + 4 + 2 * mu_eff / n)

        self._n = n
        self._lam = lam
        self._mu = mu
        self._mu_eff = mu_eff
        self._cc, self._c1, self._cmu = cc, c1, cmu
        self._c_sigma, self._d_sigma = c_sigma, d_sigma
        self._cm = 1.0
        self._chi_n = math.sqrt(n) * (1.0 - (1.0 / (4.0 * n)) + 1.0 / (21.0 * (n ** 2)))
        self._weights = weights
        self._p_sigma = np.zeros(n)
        self._pc = np.zeros(n)
        self._mean = mean.copy()
        if cov is None:
            self._C = np.eye(n)
        else:
            cov = np.asarray(cov, dtype=np.float64)
            if cov.shape != (n, n):
                raise ValueError("Invalid shape of covariance matrix")
            self._C = cov.copy()
        self._sigma = float(sigma)
        self._D = None
        self._B = None
        if bounds is not None:
            bounds = np.asarray(bounds, dtype=np.float64)
            if bounds.shape != (n, 2):
                raise ValueError("invalid bounds")
        self._bounds = bounds
        self._n_max_resampling = int(n_max_resampling)
        self._g = 0
        # PCG64 + ziggurat normals: ~4x the throughput of RandomState.randn; the reference's
        # stream is unseeded and cmaes is absent, so no particular stream has to be reproduced
        self._rng = np.random.Generator(np.random.PCG64(seed))
        # optional device sampler for ask_population(): callable(mean, sigma, BD, bounds, P, n_max_resampling,
        # seed, generation) -> (P, D); the same resample-then-clip procedure, counter-based random numbers
        self._sampler = sampler
        self._sampler_seed = int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (1 << 63))

    # ------------------------------------------------------------------ properties
    @property
    def dim(self):
        return self._n

    @property
    def population_size(self):
        return self._lam

    @property
    def generation(self):
        return self._g

    @property
    def mean(self):
        return self._mean

    @property
    def sigma(self):
        return self._sigma

    # ------------------------------------------------------------------ sampling
    def _eigen(self):
        if self._B is not None and self._D is not None:
            return self._B, self._D
        self._C = (self._C + self._C.T) / 2
        d2, b = np.linalg.eigh(self._C)
        d = np.sqrt(np.where(d2 < 0, _EPS, d2))
        self._C = np.dot(np.dot(b, np.diag(d ** 2)), b.T)
        self._B, self._D = b, d
        return b, d

    def _sample(self, count):
        b, d = self._eigen()
        z = self._rng.standard_normal((count, self._n))
        y = (z * d) @ b.T                      # rows: B diag(D) z
        return self._mean + self._sigma * y

    def _feasible(self, x):
        if self._bounds is None:
            return np.ones(x.shape[:-1], dtype=bool)
        return np.all((x >= self._bounds[:, 0]) & (x <= self._bounds[:, 1]), axis=-1)

    def _repair(self, x):
        if self._bounds is None:
            return x
        return np.clip(x, self._bounds[:, 0], self._bounds[:, 1])

    def ask(self):
        """One candidate (the reference's call pattern, optimize.py:421)."""
        for _ in range(self._n_max_resampling):
            x = self._sample(1)[0]
            if self._feasible(x):
                return x
        return self._repair(self._sample(1)[0])

    def ask_population(self):
        """All ``population_size`` candidates of one generation as a (P, D) array, with the
        same re-sample-then-clip rule applied row-wise."""
        if self._sampler is not None:
            b, d = self._eigen()
            return np.asarray(self._sampler(self._mean, self._sigma, b * d, self._bounds, self._lam,
                                            self._n_max_resampling, self._sampler_seed, self._g), dtype=np.float64)
        x = self._sample(self._lam)
        bad = np.flatnonzero(~self._feasible(x))
        tries = 1
        batch = 1
        # rows still infeasible draw their next `batch` re-samples at once (batch doubles), and
        # keep the FIRST feasible one: the same per-candidate sequence as a round-by-round loop,
        # in ~log2(n_max_resampling) numpy calls instead of n_max_resampling
        while bad.size and tries < self._n_max_resampling:
            r = min(batch, self._n_max_resampling - tries)
            cand = self._sample(bad.size * r).reshape(bad.size, r, self._n)
            ok = self._feasible(cand)                         # (bad, r)
            hit = ok.any(axis=1)
            first = ok.argmax(axis=1)
            x[bad[hit]] = cand[hit, first[hit]]
            bad = bad[~hit]
            tries += r
            batch *= 2
        if bad.size:
            x[bad] = self._repair(self._sample(bad.size))
        return x

    # ------------------------------------------------------------------ update
    def tell(self, solutions):
        """``solutions``: list of ``(x, value)``, length == population_size.  Sorted in place
        (stable, ascending value; NaN ranks last) -> ``solutions[0]`` is the generation's best."""
        if len(solutions) != self._lam:
            raise ValueError("Must tell popsize-length solutions.")
        x = np.array([s[0] for s in solutions], dtype=np.float64)
        values = np.array([s[1] for s in solutions], dtype=np.float64)
        order = self.tell_population(x, values)
        solutions[:] = [solutions[i] for i in order]

    def tell_population(self, x, values):
        """Array form of ``tell`` used by the GPU path: ``x`` (P, D) candidates, ``values`` (P,)
        losses.  Returns the stable ascending order (NaN last) that ``tell`` applies to its list."""
        x = np.asarray(x, dtype=np.float64)
        values = np.asarray(values, dtype=np.float64)
        if x.shape != (self._lam, self._n) or values.shape != (self._lam,):
            raise ValueError("Must tell popsize-length solutions.")
        if not np.all(np.abs(x) < _MEAN_MAX):
            raise ValueError(f"Abs of all param values must be less than {_MEAN_MAX} to avoid overflow errors")
        self._g += 1
        order = np.argsort(np.where(np.isnan(values), np.inf, values), kind="stable")

        b, d = self._eigen()
        self._B, self._D = None, None
        n = self._n

        x_k = x[order]
        y_k = (x_k - self._mean) / self._sigma

        y_w = np.sum(y_k[:self._mu].T * self._weights[:self._mu], axis=1)
        self._mean = self._mean + self._cm * self._sigma * y_w

        c_2 = (b / d) @ b.T                       # C^(-1/2) = B D^-1 B^T
        self._p_sigma = (1 - self._c_sigma) * self._p_sigma + math.sqrt(
            self._c_sigma * (2 - self._c_sigma) * self._mu_eff) * c_2.dot(y_w)
        norm_p_sigma = np.linalg.norm(self._p_sigma)
        self._sigma *= np.exp((self._c_sigma / self._d_sigma) * (norm_p_sigma / self._chi_n - 1))
        self._sigma = min(self._sigma, _SIGMA_MAX)

        h_left = norm_p_sigma / math.sqrt(1 - (1 - self._c_sigma) ** (2 * (self._g + 1)))
        h_right = (1.4 + 2 / (n + 1)) * self._chi_n
        h_sigma = 1.0 if h_left < h_right else 0.0

        self._pc = (1 - self._cc) * self._pc + h_sigma * math.sqrt(
            self._cc * (2 - self._cc) * self._mu_eff) * y_w

        w_io = self._weights * np.where(
            self._weights >= 0, 1, n / (np.linalg.norm(y_k @ c_2.T, axis=1) ** 2 + _EPS))
        delta_h = (1 - h_sigma) * self._cc * (2 - self._cc)
        rank_one = np.outer(self._pc, self._pc)
        rank_mu = (y_k.T * w_io) @ y_k
        self._C = ((1 + self._c1 * delta_h - self._c1 - self._cmu * np.sum(self._weights)) * self._C
                   + self._c1 * rank_one + self._cmu * rank_mu)
        return order
