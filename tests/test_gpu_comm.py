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
        finally:
            L.comm_destroy()
        again, amin2 = pts.eval_population(cand, L.LOSS_HUBER, 10.0)
    assert L.comm_info() == (0, 1)
    z = np.arange(4.0)
    L.comm_bcast(z)                                                     # no communicator: no-op
    np.testing.assert_array_equal(before, after)
    np.testing.assert_array_equal(before, again)
    assert amin0 == amin1 == amin2
