"""world_size-2 (and 3) test of the vertex-sharded population evaluation on CPU (gloo): shard
bounds, the P+1 all-reduce contract, NaN poisoning and the replicated argmin."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from alproj_amd import dist as adist
from alproj_amd import synthetic as syn
from oracle import ref_numpy as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_population_matches_single_process(tmp_path, world):
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), port, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-2000:]
    res = [np.load(o) for o in outs]
    # single-process reference over all points
    truth = syn.truth_params(316)
    init = syn.base_params(316)
    n = 3001
    xyz = syn.gcp_points(n, truth, seed=5)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(5).normal(0, 1.0, (n, 2))
    xyz[17] = [init["x"], init["y"], init["z"]]
    bounds = orc.bounds_to_array(init, syn.TARGETS_D9)
    X = np.random.default_rng(9).uniform(0.4, 0.6, (12, 9))
    X[0] = 0.5
    with np.errstate(all="ignore"):
        ref, ref_amin = orc.population_losses(xyz, uv, init, syn.TARGETS_D9, bounds, X, 10.0)
    assert np.isnan(ref[0])                             # poisoned by the at-camera point
    covered = sorted((int(r["lo"]), int(r["hi"])) for r in res)
    assert covered[0][0] == 0 and covered[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    for r in res:
        np.testing.assert_allclose(r["losses"], ref, rtol=1e-12, equal_nan=True)
        assert int(r["amin"]) == ref_amin != 0
        np.testing.assert_array_equal(r["losses"], res[0]["losses"])      # identical on every rank


def test_world8_row_shards_of_the_10000_row_grid_reproduce_the_single_process_losses():
    """BASELINE config 5's decomposition: the 10 000 grid rows of the 100 M-vertex DSM over 8 ranks
    (alproj_amd.dist.shard_rows).  On a 10 000 x 12 slice of that grid (every row, 12 columns: all the row
    bookkeeping, 1/833 of the arithmetic) the per-rank SUMS, packed and combined exactly as the all-reduce
    delivers them (pack_partials / combine_partials), equal the single-process mean losses to 1e-12, the
    argmin is the same, and every rank generates only its own rows of the surface (row-keyed noise)."""
    n_rows, n_cols, world = 10_000, 12, 8
    full = syn.dsm_rows(n_rows, 0, n_rows)[:, :]                      # (rows * n_side, 3) would be 100 M: take columns below
    # a rank's rows generated on their own equal the same rows of the whole surface
    lo3, hi3 = adist.shard_rows(n_rows, 3, world)
    np.testing.assert_array_equal(syn.dsm_rows(n_rows, lo3, hi3), full[lo3 * n_rows:hi3 * n_rows])
    grid = full.reshape(n_rows, n_rows, 3)[:, 4000:4000 + n_cols, :]
    del full
    xyz = np.ascontiguousarray(grid[:, :, [0, 2, 1]].reshape(-1, 3).astype(np.float64))      # x, y, z
    base = syn.standoff_params(n_rows)
    base = dict(base, x=base["x"] - syn.ABS_ORIGIN_XZY[0], y=base["y"] - syn.ABS_ORIGIN_XZY[2], z=base["z"] - syn.ABS_ORIGIN_XZY[1])
    truth = syn.perturbed(base)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(4).normal(0, 1.0, (len(xyz), 2))
    bounds = orc.bounds_to_array(base, syn.TARGETS_D21)
    X = np.random.default_rng(6).uniform(0.45, 0.55, (16, 21))
    ref, ref_amin = orc.population_losses(xyz, uv, base, syn.TARGETS_D21, bounds, X, 10.0)
    reduced = np.zeros(len(X) + 1)
    rows_seen = 0
    for r in range(world):
        lo, hi = adist.shard_rows(n_rows, r, world)
        rows_seen += hi - lo
        sl = slice(lo * n_cols, hi * n_cols)
        sums = np.array([orc.huber(uv[sl], orc.project_points(xyz[sl], orc.candidate_params(base, syn.TARGETS_D21, bounds, x)), 10.0)
                         * (sl.stop - sl.start) for x in X])
        reduced += adist.pack_partials(sums, sl.stop - sl.start)         # what ncclAllReduce(sum) leaves on every rank
    assert rows_seen == n_rows and reduced[-1] == len(xyz)
    losses, amin = adist.combine_partials(reduced)
    np.testing.assert_allclose(losses, ref, rtol=1e-12)
    assert amin == ref_amin


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100_000_000, 10_004_569):
        for world in (1, 2, 3, 8):
            b = [adist.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        adist.shard_bounds(10, 2, 2)


def test_combine_partials_semantics():
    red = adist.pack_partials(np.array([30.0, 10.0, np.nan, 10.0]), 10)
    losses, amin = adist.combine_partials(red)
    np.testing.assert_array_equal(losses[[0, 1, 3]], [3.0, 1.0, 1.0])
    assert np.isnan(losses[2]) and amin == 1            # first index on ties, NaN never wins
    losses, amin = adist.combine_partials(adist.pack_partials(np.array([np.nan, np.nan]), 5))
    assert amin == 0
    losses, _ = adist.combine_partials(adist.pack_partials(np.array([0.0]), 0))
    assert np.isnan(losses[0])                          # np.mean of an empty set


def _single_process_lsq(points=slice(None)):
    """LsqOptimizer.optimize of the product in THIS process over all 1201 points (world 1), the oracle standing in for the
    device exactly as in tests/_dist_cma_worker.py"""
    import pandas as pd
    from alproj_amd import _lib
    from alproj_amd import optimize as aopt
    truth = syn.truth_params(316)
    init = dict(truth, pan=truth["pan"] + 1.5, tilt=truth["tilt"] - 1.0, fov=truth["fov"] + 2, x=truth["x"] + 3)
    n = 1201
    xyz = syn.gcp_points(n, truth, seed=11)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(11).normal(0, 0.8, (n, 2))
    xyz, uv = xyz[points], uv[points]

    class AllPoints:
        precision, n = _lib.ALP_F64, len(xyz)

        def eval_population(self, cand, kind, f_scale, want_argmin=True):
            return np.array([orc.mean_distance(uv, orc.project_points(xyz, orc.vector_to_params(c))) for c in cand]), 0

        def residuals(self, vec):
            return orc.residual_vector(xyz, uv, orc.vector_to_params(vec))

        def residuals_batch(self, cand):
            return np.stack([self.residuals(c) for c in cand])

        def close(self):
            pass

    saved = aopt.BaseOptimizer._device_points, _lib.comm_info
    aopt.BaseOptimizer._device_points = lambda self, precision: AllPoints()
    _lib.comm_info = lambda: (0, 1)
    out = {}
    try:
        obj, img = pd.DataFrame(xyz, columns=["x", "y", "z"]), pd.DataFrame(uv, columns=["u", "v"])
        for tag, kw in (("lsq", dict(method="trf", loss="linear", max_nfev=30)),
                        ("lsq_huber", dict(method="trf", loss="huber", f_scale=2.0, max_nfev=30)),
                        ("lsq_2point", dict(method="dogbox", loss="linear", jac="2-point", max_nfev=30))):
            q = aopt.LsqOptimizer(obj, img, init)
            q.set_target(["fov", "pan", "tilt", "roll"])
            lp, lerr = q.optimize(**kw)
            out[tag + "_params"] = np.array([lp[k] for k in _lib.PARAM_KEYS], dtype=np.float64)
            out[tag + "_err"] = lerr
    finally:
        aopt.BaseOptimizer._device_points, _lib.comm_info = saved
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_optimisers_multi_rank_branches_stay_in_lockstep(tmp_path, world):
    """The multi-rank branches of CMAOptimizer.optimize (seed broadcast before the sampler exists, candidate
    matrix broadcast every generation, final error as a collective) and of LsqOptimizer.optimize (rank 0's
    solution broadcast before the final collective) run for real in `world` processes over gloo, each on its
    shard (tests/_dist_cma_worker.py).  Nobody passes a seed, as in the reference: unless rank 0's entropy and
    candidates reach every rank, the ranks sample different populations and the all-reduce adds sums of
    different candidates -- here every rank must end with bit-identical candidates for 20 generations,
    parameters and error."""
    port = _free_port()
    outs = [str(tmp_path / f"c{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_cma_worker.py"), str(r), str(world), port, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0, out.decode()[-3000:]
    res = [np.load(o) for o in outs]
    for r in res[1:]:
        np.testing.assert_array_equal(r["cma_X"], res[0]["cma_X"])
        np.testing.assert_array_equal(r["cma_params"], res[0]["cma_params"])
        assert float(r["cma_err"]) == float(res[0]["cma_err"])
        for tag in ("lsq", "lsq_huber", "lsq_2point"):
            np.testing.assert_array_equal(r[tag + "_params"], res[0][tag + "_params"])
            assert float(r[tag + "_err"]) == float(res[0][tag + "_err"])
    assert res[0]["cma_X"].shape == (20, 12, 9)
    # the order of collectives every rank went through: seed, then (candidates, evaluation) x 20, then the final error
    expect = ["('bcast', 'uint64', (2,))"] + ["('bcast', 'float64', (12, 9))", "('eval', 12)"] * 20 + ["('eval', 1)"]
    for r in res:
        assert list(r["cma_log"]) == expect
        log = list(r["lsq_log"])
        assert log[-2:] == ["('bcast', 'float64', (4,))", "('eval', 1)"]
        assert len(log) > 4 and all(e.startswith("('allgather'") for e in log[:-2])       # residual vectors (1-D) and Jacobian rows (., 4)
        assert any(e.endswith(", 4))") for e in log[:-2]) and not any(e.endswith(", 4))") for e in r["lsq_2point_log"])
    # the reference's problem, not a shard's: the single-process solve over ALL points (same host code, the oracle as the
    # device) gives the same optimum, to the tolerances of g14 (tests/test_gpu_golden_render.py: what the reference's own
    # optimum moves by when its residuals change in the last bits -- numpy's BLAS rounds a shard's np.dot differently from
    # the whole array's, and 2-point differences with a 1.5e-8 step amplify that): 2e-4 degrees, 5e-5 px on the error.
    # A shard-only solve (the round-3 behaviour) is 1e-2 ... 1e-1 degrees away.
    single = _single_process_lsq()
    for tag in ("lsq", "lsq_huber", "lsq_2point"):
        np.testing.assert_allclose(res[0][tag + "_params"], single[tag + "_params"], rtol=0, atol=2e-4)
        assert abs(float(res[0][tag + "_err"]) - single[tag + "_err"]) < 5e-5
    shard = _single_process_lsq(points=slice(0, 1201 // world))
    assert np.abs(shard["lsq_params"] - single["lsq_params"]).max() > 2e-3
    # ... and least squares found the pose (20 generations of 12 do not finish CMA-ES's 9-parameter search, and
    # its result is the last generation's best, quirk Q9: nothing to assert on its error but that it is finite)
    assert np.isfinite(float(res[0]["cma_err"])) and float(res[0]["lsq_err"]) < 10.0 < float(res[0]["init_err"])
