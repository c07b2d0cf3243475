"""RCCL path of the library on ONE GPU: a world_size-1 communicator still runs
ncclCommInitRank and routes the P+1 partial sums through ncclAllReduce on the library stream
(8-GPU runs are the driver's; rank > 0 cannot be created on a 1-GPU box)."""
import numpy as np
import pytest

from oracle import ref_numpy as orc

pytestmark = pytest.mark.gpu


def test_allreduce_path_world1():
    from alproj_amd import _lib as L
    from alproj_amd import synthetic as syn
    L.init(0)
    truth = syn.truth_params(316)
    init = syn.base_params(316)
    xyz = syn.gcp_points(3000, truth, seed=2)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(2).normal(0, 1, (3000, 2))
    cand = np.tile(L.params_vector(init), (300, 1))
    cand[:, 4] += np.random.default_rng(3).uniform(-3, 3, 300)          # pan
    with L.Points(xyz, [init["x"], init["y"], init["z"]], "f64") as pts:
        pts.set_observed(uv)
        before, amin0 = pts.eval_population(cand, L.LOSS_HUBER, 10.0)
        uid = L.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        L.comm_init(uid, 0, 1)
        try:
            with pytest.raises(L.AlprojHipError):
                L.comm_init(uid, 0, 1)                                   # already exists
            after, amin1 = pts.eval_population(cand, L.LOSS_HUBER, 10.0)
            assert L.comm_info() == (0, 1)
            seed = np.array([0x1234567890ABCDEF], dtype=np.uint64)
            L.comm_bcast(seed, root=0)                                   # ncclBroadcast through the library stream
            assert int(seed[0]) == 0x1234567890ABCDEF
            X = np.random.default_rng(0).random((50, 21))
            Y = X.copy()
            L.comm_bcast(Y)
            np.testing.assert_array_equal(X, Y)
            # all-gather of ragged host arrays (LsqOptimizer's residual vectors and Jacobian rows): ncclAllGather of the
            # counts, one ncclBroadcast per rank
            np.testing.assert_array_equal(L.comm_allgather(X), X)
            r = np.random.default_rng(1).random(1601)
            np.testing.assert_array_equal(L.comm_allgather(r), r)
            assert L.comm_allgather(X[:0]).shape == (0, 21)
        finally:
            L.comm_destroy()
        again, amin2 = pts.eval_population(cand, L.LOSS_HUBER, 10.0)
    assert L.comm_info() == (0, 1)
    z = np.arange(4.0)
    L.comm_bcast(z)                                                     # no communicator: no-op
    np.testing.assert_array_equal(L.comm_allgather(z), z)               # ... and a copy
    assert len(L.device_info()["pci_bus_id"]) >= 7
    np.testing.assert_array_equal(before, after)
    np.testing.assert_array_equal(before, again)
    assert amin0 == amin1 == amin2


def test_optimisers_take_rank0s_seed_and_candidates_on_the_device(monkeypatch):
    """The multi-rank branch of CMAOptimizer.optimize (optimize.py of the product, reference loop :410-427) with the
    REAL device evaluation: `_lib.comm_info` says rank 1 of 2 and `_lib.comm_bcast` is a stand-in for rank 0 that
    overwrites what it is given -- the seed, and every generation's candidate matrix -- and records the order of
    events.  The optimiser must build its sampler from the broadcast seed, evaluate and `tell` exactly the broadcast
    candidates (not its own), and still reach the final error; LsqOptimizer must return the broadcast solution."""
    import pandas as pd
    from alproj_amd import _lib as L
    from alproj_amd import optimize as aopt
    from alproj_amd import synthetic as syn
    L.init(0)
    truth = syn.truth_params(316)
    init = dict(truth, pan=truth["pan"] + 1.5, tilt=truth["tilt"] - 1.0, fov=truth["fov"] + 2)
    xyz = syn.gcp_points(800, truth, seed=4)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(4).normal(0, 0.8, (800, 2))
    obj, img = pd.DataFrame(xyz, columns=["x", "y", "z"]), pd.DataFrame(uv, columns=["u", "v"])
    events, sent = [], []
    rank0 = np.random.default_rng(99)

    def fake_bcast(array, root=0):
        assert root == 0 and array.flags["C_CONTIGUOUS"]
        if array.dtype == np.uint64:
            array[...] = (424242, 1)                                    # "rank 0's" seed and element type (1 = float64)
        else:
            array[...] = rank0.uniform(0.3, 0.7, array.shape)           # "rank 0's" candidates / solution
            sent.append(array.copy())
        events.append(("bcast", str(array.dtype), array.shape))
        return array

    def fake_allgather(array):                                          # "rank 0" holds no points: the gathered rows are this rank's
        events.append(("allgather", array.shape))
        return np.ascontiguousarray(array)

    monkeypatch.setattr(L, "comm_info", lambda: (1, 2))
    monkeypatch.setattr(L, "comm_bcast", fake_bcast)
    monkeypatch.setattr(L, "comm_allgather", fake_allgather)
    real_cma = aopt.CMA

    class SpyCMA(real_cma):
        def __init__(self, *a, **k):
            events.append(("cma", k.get("seed")))
            super().__init__(*a, **k)

        def tell_population(self, X, losses):
            events.append(("tell", np.array(X, copy=True), np.array(losses, copy=True)))
            return super().tell_population(X, losses)

    monkeypatch.setattr(aopt, "CMA", SpyCMA)
    real_eval = L.Points.eval_population

    def spy_eval(self, cand, kind, f_scale=10.0, want_argmin=True):
        events.append(("eval", np.array(cand, copy=True), want_argmin, self.precision))
        return real_eval(self, cand, kind, f_scale, want_argmin)

    monkeypatch.setattr(L.Points, "eval_population", spy_eval)
    o = aopt.CMAOptimizer(obj, img, init)
    o.set_target(["fov", "pan", "tilt", "roll"])
    gens, pop = 6, 10
    params, err = o.optimize(generation=gens, sigma=0.3, population_size=pop, f_scale=10.0, seed=None, progress=False, precision="f32")
    kinds = [e[0] for e in events]
    assert kinds == ["bcast", "cma"] + ["bcast", "eval", "tell"] * gens + ["eval"]
    # the argmin (with its float64 confirmation on float32 sets) is asked for in the LAST generation only (optimize.py:427)
    assert [e[2] for e in events if e[0] == "eval"] == [False] * (gens - 1) + [True, True]
    assert events[0][1:] == ("uint64", (2,)) and events[1][1] == 424242           # the sampler is built from rank 0's seed
    # ... and the point set has rank 0's element type, not the one this rank asked for: the float64 confirmation of a
    # float32 set is a collective of its own, so the ranks must not differ
    assert all(e[3] == L.ALP_F64 for e in events if e[0] == "eval")
    bounds = aopt.bounds_to_array(init, o.target_params)
    cols = [L.PARAM_KEYS.index(t) for t in o.target_params]
    for g in range(gens):
        b, e, t = events[2 + 3 * g: 5 + 3 * g]
        X = sent[g]
        assert b[2] == (pop, 4)
        np.testing.assert_array_equal(t[1], X)                                    # told what was broadcast ...
        np.testing.assert_array_equal(e[1][:, cols], X * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0])   # ... and evaluated it
        ref, ref_amin = orc.population_losses(xyz, uv, init, o.target_params, bounds, X, 10.0)
        np.testing.assert_allclose(t[2], ref, rtol=1e-9)
    # the result is the last generation's best candidate (quirk Q9) of the BROADCAST matrix; the final error was reached
    last = sent[gens - 1]
    best = last[ref_amin] * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]
    np.testing.assert_array_equal([params[t] for t in o.target_params], best)
    assert events[-1][1].shape == (1, 25) and np.isfinite(err)
    # least squares: rank 0's solution is the result
    del events[:], sent[:]
    q = aopt.LsqOptimizer(obj, img, init)
    q.set_target(["fov", "pan", "tilt", "roll"])
    lp, lerr = q.optimize(method="trf", loss="linear", max_nfev=20)
    gathers = [e for e in events if e[0] == "allgather"]
    events[:] = [e for e in events if e[0] != "allgather"]
    assert {g[1] for g in gathers} == {(1600,), (1600, 4)}             # every residual vector and every Jacobian went through it
    assert [e[0] for e in events] == ["bcast", "eval"] and events[0][1:] == ("float64", (4,))
    np.testing.assert_array_equal([lp[t] for t in q.target_params], sent[0])
    np.testing.assert_array_equal(events[1][1][0, cols], sent[0])
    assert np.isfinite(lerr)
