"""CPU tests of the host-side mirror of the reference interface: the CMA-ES sampler contract,
bounds / candidate-matrix construction, the small matrix helpers (against the golden vectors
generated from the reference) and argument validation.  No GPU work."""
import os

import numpy as np
import pandas as pd
import pytest

from alproj_amd import optimize as opt
from alproj_amd import project as prj
from alproj_amd.cma import CMA
from oracle import ref_numpy as orc

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ------------------------------------------------------------------ matrices and bounds
def test_intrinsic_extrinsic_golden():
    g = load("g1_matrices.npz")
    for pv, K, E in zip(g["params"], g["K"], g["E"]):
        p = orc.vector_to_params(pv)
        np.testing.assert_allclose(opt.intrinsic_mat(p["fov"], p["w"], p["h"], p["cx"], p["cy"]), K, rtol=1e-15)
        np.testing.assert_allclose(opt.extrinsic_mat(p["pan"], p["tilt"], p["roll"], p["x"], p["y"], p["z"]),
                                   E, rtol=1e-13, atol=1e-9)
    np.testing.assert_allclose(opt.intrinsic_mat(75, 5616, 3744), g["K_default_75_5616_3744"], rtol=1e-15)


def test_gl_matrices_golden():
    g = load("g7_gl_matrices.npz")
    off = g["cam_offset"]
    for pv, pm, mv in zip(g["params"], g["proj"], g["view"]):
        p = orc.vector_to_params(pv)
        np.testing.assert_array_equal(prj.projection_mat(p["fov"], p["w"], p["h"]), pm)
        np.testing.assert_allclose(prj.modelview_mat(p["pan"], p["tilt"], p["roll"], p["x"] - off[0],
                                                     p["y"] - off[1], p["z"] - off[2]), mv, rtol=1e-13, atol=1e-9)
    np.testing.assert_array_equal(prj.projection_mat(75, 5616, 3744, near=0.5, far=5000.0, cx=2800.0, cy=1880.0),
                                  g["proj_cxcy_near_far"])
    # quirk Q10: defaults near=-1, far=1 give rows 3/4 = [0,0,0,1], [0,0,-1,0]
    pm = prj.projection_mat(75, 5616, 3744)
    np.testing.assert_allclose(pm[8:], [0, 0, 0, 1, 0, 0, -1, 0], atol=1e-15)
    assert abs(pm[0] - 1.303225) < 1e-6 and abs(pm[5] - 2.144507) < 1e-6


def test_bounds_to_array_golden():
    g = load("g6_bounds.npz")
    p = orc.vector_to_params(g["params"])
    tgt = [str(t) for t in g["targets"]]
    np.testing.assert_array_equal(opt.bounds_to_array(p, tgt), g["default"])
    np.testing.assert_array_equal(opt.bounds_to_array(p, tgt, {"fov": 10, "cx": 7.5}), g["override"])
    np.testing.assert_array_equal(opt.bounds_to_array(p, [str(t) for t in g["all21_targets"]]), g["all21"])
    assert opt.DEFAULT_BOUND_WIDTHS == orc.DEFAULT_BOUND_WIDTHS


def test_set_target_and_candidate_matrix():
    g = load("g5_population.npz")
    init = orc.vector_to_params(g["params_init"])
    o = opt.CMAOptimizer(pd.DataFrame(g["xyz"], columns=["x", "y", "z"]),
                         pd.DataFrame(g["uv_obs"], columns=["u", "v"]), init)
    o.set_target()
    assert o.target_params == ["fov", "pan", "tilt", "roll", "a1", "a2", "k1", "k2", "k3", "k4", "k5", "k6",
                               "p1", "p2", "s1", "s2", "s3", "s4"]                     # x, y, z excluded
    tgt = [str(t) for t in g["d9_targets"]]
    o.set_target(tgt)
    np.testing.assert_array_equal(o.target_params_init, [init[t] for t in tgt])
    bounds = g["d9_bounds"]
    X = g["d9_X"]
    cand = o._candidate_matrix(X * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0])
    assert cand.shape == (len(X), 25)
    for i in (0, 5, 23):                     # row i == the dict the reference's closure builds
        ref = orc.candidate_params(init, tgt, bounds, X[i])
        np.testing.assert_array_equal(cand[i], orc.params_to_vector(ref))
    res = o._result_params(cand[3][[orc.PARAM_KEYS.index(t) for t in tgt]])
    assert set(res) == set(init) and res["w"] == init["w"]


def test_default_precision_is_the_references_float64_at_gcp_scale(monkeypatch):
    """CMAOptimizer.optimize(precision=None): a float64 point set up to F64_MAX_POINTS points (the reference's arithmetic,
    optimize.py:329-357), float32 above; an explicit choice is kept.  The device is a recorder here: no GPU."""
    from alproj_amd import _lib
    assert opt.default_precision(1127) == "f64" and opt.default_precision(opt.F64_MAX_POINTS) == "f64"
    assert opt.default_precision(opt.F64_MAX_POINTS + 1) == "f32" and opt.default_precision(100_000_000) == "f32"
    assert opt.default_precision(100_000_000, "f64") == "f64" and opt.default_precision(10, "f32") == "f32"
    with pytest.raises(ValueError):
        opt.default_precision(10, "f16")
    made = []

    class Recorder:
        def __init__(self, xyz, origin, precision="f32"):
            made.append((len(xyz), precision))
            self.precision, self.n = (_lib.ALP_F64 if precision == "f64" else _lib.ALP_F32), len(xyz)

        @classmethod
        def from_columns(cls, x, y, z, origin, precision="f32"):
            return cls(x, origin, precision)

        def set_observed(self, uv):
            pass

        set_observed_columns = lambda self, u, v: None

        def eval_population(self, cand, kind, f_scale, want_argmin=True):
            return np.arange(len(cand), dtype=np.float64), 0

        def close(self):
            pass

        __enter__ = lambda self: self
        __exit__ = lambda self, *a: None

    monkeypatch.setattr(_lib, "Points", Recorder)
    monkeypatch.setattr(_lib, "comm_info", lambda: (0, 1))
    g = load("g5_population.npz")
    init = orc.vector_to_params(g["params_init"])
    o = opt.CMAOptimizer(pd.DataFrame(g["xyz"], columns=["x", "y", "z"]), pd.DataFrame(g["uv_obs"], columns=["u", "v"]), init)
    o.set_target(["pan", "tilt"])
    o.optimize(generation=2, population_size=4, progress=False, seed=1)
    assert made == [(len(g["xyz"]), "f64")]
    del made[:]
    o.optimize(generation=2, population_size=4, progress=False, seed=1, precision="f32")
    assert made == [(len(g["xyz"]), "f32"), (len(g["xyz"]), "f64")]       # explicit float32; the final error from a float64 copy
    del made[:]
    monkeypatch.setattr(opt, "F64_MAX_POINTS", 100)                        # a "DSM-sized" set without allocating one
    monkeypatch.setattr(opt.CMAOptimizer, "F64_FINAL_MAX_POINTS", 100)
    o.optimize(generation=2, population_size=4, progress=False, seed=1)
    assert made == [(len(g["xyz"]), "f32")]


def test_table_columns_reach_the_upload_without_host_copies():
    """optimize._columns: a frame of exactly the float64 columns hands out its block (row-major, or the transposed view of a
    columns x rows block -- both uploaded as they lie); other frames give their columns one by one.  No GPU."""
    rng = np.random.default_rng(0)
    a = rng.random((1000, 3))
    f1 = pd.DataFrame(a, columns=["x", "y", "z"])
    g1 = opt._xyz_array(f1)
    assert isinstance(g1, np.ndarray) and g1.flags["C_CONTIGUOUS"] and np.shares_memory(g1, f1.to_numpy(copy=False))
    f2 = pd.DataFrame({"x": a[:, 0].copy(), "y": a[:, 1].copy(), "z": a[:, 2].copy()})
    g2 = opt._xyz_array(f2)
    assert isinstance(g2, list) and len(g2) == 3 and all(c.flags["C_CONTIGUOUS"] and np.shares_memory(c, f2[k].to_numpy()) for c, k in zip(g2, "xyz"))
    np.testing.assert_array_equal(opt._as_rows(g2), a)
    u, v = a[:, 0].copy(), a[:, 1].copy()
    f5 = pd.DataFrame({"u": u, "v": v}, copy=False)                 # what project() returns: one block per column
    g5 = opt._uv_array(f5)
    assert isinstance(g5, list) and np.shares_memory(g5[0], u) and np.shares_memory(g5[1], v)
    f3 = pd.DataFrame({"id": np.arange(1000), "z": a[:, 2], "x": a[:, 0], "y": a[:, 1]})
    g3 = opt._xyz_array(f3)
    assert isinstance(g3, list) and opt._rows(g3) == 1000
    np.testing.assert_array_equal(opt._as_rows(g3), a)
    f4 = pd.DataFrame({"u": a[:, 0].astype(np.float32), "v": a[:, 1]})          # mixed types: per column, as float64
    g4 = opt._uv_array(f4)
    assert isinstance(g4, list) and all(c.dtype == np.float64 for c in g4)
    assert opt._rows(np.zeros((7, 3))) == 7 and opt._xyz_array(a) is a


def test_init_comm_and_the_file_rendezvous_without_a_gpu(monkeypatch, tmp_path):
    """alproj_amd.dist.init_comm / init_from_file with the library stood in for: rank 0's 128 bytes reach every rank, every rank
    initialises its device first, and a rank that SEES one device although LOCAL_RANK >= 1 (a launcher isolating ranks by
    HIP_VISIBLE_DEVICES) takes device 0"""
    import threading
    from alproj_amd import _lib
    from alproj_amd import dist as adist
    log = []
    monkeypatch.setattr(_lib, "init", lambda device=None: log.append(("init", device)))
    monkeypatch.setattr(_lib, "comm_unique_id", lambda: bytes(range(128)))
    monkeypatch.setattr(_lib, "comm_init", lambda uid, rank, world: log.append(("comm_init", uid, rank, world)))
    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    adist.init_comm(0, 1, None, device=0)                         # one rank: no id, no communicator
    assert log == [("init", 0)]
    del log[:]
    adist.init_comm(0, 2, lambda b: b, device=0)
    adist.init_comm(1, 2, lambda b: bytes(range(128)), device=1)
    assert log == [("init", 0), ("comm_init", bytes(range(128)), 0, 2), ("init", 1), ("comm_init", bytes(range(128)), 1, 2)]
    del log[:]
    monkeypatch.setattr(_lib, "device_count", lambda: 1)          # isolated by visibility
    adist.init_comm(3, 8, lambda b: bytes(range(128)), device=3)
    assert log[0] == ("init", 0) and log[1][2:] == (3, 8)
    del log[:]
    # the file rendezvous: rank 1 polls until rank 0 has written
    path = str(tmp_path / "uid")
    got = {}

    def rank1():
        got["r1"] = adist.init_from_file(path, 1, 2, device=1, timeout_s=20.0)

    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    th = threading.Thread(target=rank1)
    th.start()
    assert adist.init_from_file(path, 0, 2, device=0) == (0, 2)
    th.join(20)
    assert got["r1"] == (1, 2) and sorted(e for e in log if e[0] == "comm_init") == [("comm_init", bytes(range(128)), 0, 2), ("comm_init", bytes(range(128)), 1, 2)]
    with pytest.raises(TimeoutError):
        adist.init_from_file(str(tmp_path / "never"), 1, 2, device=1, timeout_s=0.2)
    assert adist.shard_bounds(10, 0, 3) == (0, 3) and adist.shard_rows(10, 2, 3) == (6, 10)
    with pytest.raises(ValueError):
        adist.shard_bounds(10, 3, 3)


def test_pixel_tables_of_the_wrong_shape_are_refused():
    with pytest.raises(ValueError, match="shape"):
        opt._uv_pointers(np.zeros((5, 3)))


def test_lsq_argument_errors_need_no_gpu():
    o = opt.LsqOptimizer(pd.DataFrame(np.zeros((3, 3)), columns=["x", "y", "z"]),
                         pd.DataFrame(np.zeros((3, 2)), columns=["u", "v"]), {k: 1.0 for k in orc.PARAM_KEYS})
    o.set_target(["pan"])
    with pytest.raises(ValueError, match="does not support bounds"):
        o.optimize(method="lm", bound_widths={"pan": 1})
    with pytest.raises(ValueError, match="robust loss"):
        o.optimize(method="lm", loss="huber")


def test_reverse_proj_channel_check_needs_no_gpu():
    with pytest.raises(ValueError, match="channels"):
        prj.reverse_proj(np.zeros((4, 4, 3)), None, None, {}, chnames=["a"])


# ------------------------------------------------------------------ CMA-ES sampler contract
def test_cma_defaults_and_ask():
    c = CMA(mean=np.full(9, 0.5), sigma=0.2, seed=1)
    assert c.population_size == 4 + int(3 * np.log(9)) and c.dim == 9 and c.generation == 0
    b = np.column_stack([np.zeros(9), np.ones(9)])
    c = CMA(mean=np.full(9, 0.5), sigma=1.0, bounds=b, population_size=50, n_max_resampling=100, seed=2)
    xs = np.array([c.ask() for _ in range(200)])
    assert xs.shape == (200, 9) and (xs >= 0).all() and (xs <= 1).all()
    X = c.ask_population()
    assert X.shape == (50, 9) and (X >= 0).all() and (X <= 1).all()
    # resample-then-clip: with a huge sigma and a single resampling nearly everything is clipped
    c2 = CMA(mean=np.full(4, 0.5), sigma=50.0, bounds=np.column_stack([np.zeros(4), np.ones(4)]),
             population_size=64, n_max_resampling=1, seed=3)
    Y = c2.ask_population()
    assert (Y >= 0).all() and (Y <= 1).all() and ((Y == 0) | (Y == 1)).mean() > 0.8


def test_cma_tell_sorts_in_place_stably():
    c = CMA(mean=np.zeros(3), sigma=1.0, population_size=6, seed=4)
    X = c.ask_population()
    vals = [5.0, 1.0, 3.0, 1.0, float("nan"), 2.0]
    sol = [(X[i], vals[i]) for i in range(6)]
    c.tell(sol)
    assert [s[1] for s in sol[:5]] == [1.0, 1.0, 2.0, 3.0, 5.0] and np.isnan(sol[5][1])
    assert sol[0][0] is X[1] or np.array_equal(sol[0][0], X[1])       # first of the tie (Q9)
    assert c.generation == 1
    with pytest.raises(ValueError):
        c.tell(sol[:3])


def test_cma_reproducible_and_converges():
    def run(seed, f, d, gens, pop):
        c = CMA(mean=np.full(d, 0.8), sigma=0.3, bounds=np.column_stack([np.full(d, -2.0), np.full(d, 2.0)]),
                population_size=pop, seed=seed)
        best = np.inf
        for _ in range(gens):
            X = c.ask_population()
            v = [f(x) for x in X]
            best = min(best, min(v))
            c.tell([(X[i], v[i]) for i in range(pop)])
        return best, c.mean.copy()

    sphere = lambda x: float(np.sum((x - 0.3) ** 2))
    b1, m1 = run(7, sphere, 9, 120, 16)
    b2, m2 = run(7, sphere, 9, 120, 16)
    assert b1 == b2 and np.array_equal(m1, m2)                      # same seed -> same trajectory
    assert b1 < 1e-10 and np.allclose(m1, 0.3, atol=1e-4)
    rosen = lambda x: float(np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))
    b3, m3 = run(11, rosen, 5, 400, 24)
    assert b3 < 1e-8 and np.allclose(m3, 1.0, atol=1e-3)


def test_cma_argument_checks():
    with pytest.raises(ValueError):
        CMA(mean=np.zeros(3), sigma=0.0)
    with pytest.raises(ValueError):
        CMA(mean=np.zeros(3), sigma=1.0, bounds=np.zeros((2, 2)))
    with pytest.raises(ValueError):
        CMA(mean=np.zeros(3), sigma=1.0, population_size=0)


def test_resident_mesh_cache_keys():
    """alproj_amd.project's opt-in mesh cache (set_mesh_cache): the key is the array object + layout + a digest of EVERY byte
    (alp_host_hash64), or identity alone for arrays nobody can write; the advisor's three round-3 counter-examples -- a
    contiguous in-place edit of 1023 rows between two sampled rows, of colour rows 1025..2047, of one index row -- are
    all seen"""
    import gc
    import numpy as np
    from alproj_amd import project as aproj
    rng = np.random.default_rng(0)
    vert = rng.random((50_000, 3))
    key = aproj._key(vert)
    assert aproj._same(key, vert)
    assert not aproj._same(key, vert.copy())                      # other object, same content: another mesh
    assert not aproj._same(key, None) and aproj._same(None, None) and not aproj._same(None, vert)
    vert[0, 0] += 1.0                                              # head row
    assert not aproj._same(key, vert)
    vert[0, 0] -= 1.0
    assert aproj._same(key, vert)
    vert[2048, 1] = -5.0                                           # a row the round-3 fingerprint sampled
    assert not aproj._same(key, vert)
    key = aproj._key(vert)
    vert[5121:6144, 1] += 100                                      # ... and 1023 rows it did not
    assert not aproj._same(key, vert)
    key = aproj._key(vert)
    vert[31_337, 2] = np.nextafter(vert[31_337, 2], 2.0)           # one bit of one coordinate
    assert not aproj._same(key, vert)
    key = aproj._key(vert)
    vert += 1e-9                                                   # a global edit
    assert not aproj._same(key, vert)
    key = aproj._key(vert)
    view = vert[::2]
    assert not aproj._same(key, view)                              # another layout
    ind = np.arange(30, dtype=np.int64).reshape(10, 3)
    k2 = aproj._key(ind)
    assert aproj._same(k2, ind) and aproj._same(aproj._key(np.zeros((0, 3), np.int64)), None) is False
    big = np.arange(300_000, dtype=np.int64).reshape(-1, 3)
    k3 = aproj._key(big)
    big[70_001] = 0                                                # one index row
    assert not aproj._same(k3, big)
    # read-only all the way down: identity is enough (no digest is taken), until somebody makes it writeable again
    frozen = rng.random((1000, 3))
    frozen.setflags(write=False)
    kf = aproj._key(frozen)
    assert kf[3] == "sample" and aproj._same(kf, frozen) and aproj._immutable(frozen)
    frozen.setflags(write=True)
    assert not aproj._same(kf, frozen)
    # flags toggled, array rewritten, flags toggled back -- same object, read-only again, other content
    frozen.setflags(write=False)
    assert aproj._same(kf, frozen)
    frozen.setflags(write=True)
    frozen[500, 1] += 1.0
    frozen.setflags(write=False)
    assert not aproj._same(kf, frozen)
    # ... and on an array above the full-digest size: a rewrite is seen through the sampled blocks
    old_full = aproj._SAMPLE_FULL_BELOW
    aproj._SAMPLE_FULL_BELOW = 1 << 20
    try:
        large = rng.random((3_000_000, 3))                           # 72 MB
        large.setflags(write=False)
        kl = aproj._key(large)
        assert kl[3] == "sample" and aproj._same(kl, large)
        large.setflags(write=True)
        large *= 1.0000001
        large.setflags(write=False)
        assert not aproj._same(kl, large)
        kl = aproj._key(large)
        large.setflags(write=True)
        large[-1, 2] = -1.0                                          # the tail block is always sampled
        large.setflags(write=False)
        assert not aproj._same(kl, large)
    finally:
        aproj._SAMPLE_FULL_BELOW = old_full
    frozen.setflags(write=True)
    ro_view = frozen[:]
    ro_view.setflags(write=False)
    assert not aproj._immutable(ro_view)                           # its base can still be written
    other = view.copy()
    del vert, view
    gc.collect()
    assert key[0]() is None and not aproj._same(key, other)        # the array is gone: the weak reference says so


def test_result_memory_is_recycled_only_when_nobody_holds_it():
    """_lib.result_empty: a large result's memory goes back to the pool when the array AND every view of it are gone, the next
    result of that size is written into it (touched pages: the device -> host copy runs 4-7 x faster into them); while anything
    still refers to the memory it is never handed out again; small results, a pool of size 0 and results beyond the cap are
    plain np.empty"""
    import gc
    from alproj_amd import _lib as L
    old_cap = L._pool_cap
    try:
        L.set_result_pool(0)
        L.set_result_pool(64 << 20)
        a = L.result_empty((3, 2000, 2000), np.uint8)
        assert a.shape == (3, 2000, 2000) and a.dtype == np.uint8 and a.flags.writeable and a.flags.c_contiguous
        a[:] = 7
        where = a.ctypes.data
        row = a[1, 5]                                  # a view of a view
        frame = pd.DataFrame({"v": a.reshape(-1)[:1000]}, copy=False)
        del a
        gc.collect()
        b = L.result_empty((3, 2000, 2000), np.uint8)
        assert b.ctypes.data != where                  # `row` still sees the first array's memory
        assert (row == 7).all()
        del row, frame
        gc.collect()
        c = L.result_empty((12_000_000,), np.uint8)    # same size in bytes: the first array's memory, whatever the shape
        assert c.ctypes.data == where
        f = L.result_empty((1_500_000,), np.float64)   # 12 MB as float64
        assert f.dtype == np.float64 and f.flags.aligned
        del b, c, f
        gc.collect()
        assert L._pool_bytes == 2 * 12_000_000           # three came back: no size keeps more than two
        small = L.result_empty((1000, 3), np.float64)
        assert small.flags.owndata
        big = L.result_empty((65 << 20,), np.uint8)    # beyond the cap: never kept
        assert big.flags.owndata
        L.set_result_pool(0)
        assert L._pool_bytes == 0 and L.result_empty((3, 2000, 2000), np.uint8).flags.owndata
    finally:
        L.set_result_pool(0)
        L.set_result_pool(old_cap)


def test_result_pool_evicts_other_sizes_first_and_can_be_cleared():
    """a workload that changes its raster size must not pin the old size's buffers (and recycle nothing):
    over the cap the OLDEST buffers of OTHER sizes go first; clear_result_pool() gives everything back."""
    import gc
    from alproj_amd import _lib as L
    old_cap = L._pool_cap
    try:
        L.set_result_pool(0)
        L.set_result_pool(40 << 20)
        ev0 = L.POOL_STATS["evicted"]
        a1, a2 = L.result_empty((16 << 20,), np.uint8), L.result_empty((16 << 20,), np.uint8)       # the "earlier resolution"
        del a1, a2
        gc.collect()
        assert L._pool_bytes == 32 << 20 and sorted(L._pool) == [16 << 20]
        b = L.result_empty((20 << 20,), np.uint8)                                                     # the new raster size
        where = b.ctypes.data
        del b
        gc.collect()
        # 32 + 20 > 40: one 16 MB buffer (the older one) went, the 20 MB one is kept and is recycled
        assert L._pool_bytes == (16 << 20) + (20 << 20) and L.POOL_STATS["evicted"] == ev0 + 1
        b2 = L.result_empty((20 << 20,), np.uint8)
        assert b2.ctypes.data == where
        c = L.result_empty((20 << 20,), np.uint8)
        del b2, c
        gc.collect()
        # two of 20 MB = 40 MB: the last 16 MB buffer went as well, nothing of the old size stays resident
        assert sorted(L._pool) == [20 << 20] and L._pool_bytes == 40 << 20 and len(L._pool_age) == 2
        L.set_result_pool(25 << 20)                        # a smaller cap releases the oldest
        assert L._pool_bytes == 20 << 20
        L.clear_result_pool()
        assert L._pool_bytes == 0 and not L._pool and not L._pool_age
        d = L.result_empty((20 << 20,), np.uint8)
        assert d.ctypes.data is not None and L._pool_cap == 25 << 20      # the pool is still on
    finally:
        L.set_result_pool(0)
        L.set_result_pool(old_cap)


def test_host_minmax_is_numpys_min_and_max():
    """alp_host_minmax (to_geotiff's x.min(), x.max(), y.min(), y.max(), project.py:420-423, in one threaded pass per column):
    numpy's values for any length and thread count, NaN for both when any value is NaN, -0.0 / +0.0 as equal as numpy has them"""
    from alproj_amd import _lib as L
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 1000, (1 << 20) + 1, 5_000_001):
        a = rng.normal(0, 1e3, n)
        for threads in (0, 1, 3, 16):
            lo, hi = L.host_minmax(a, threads)
            assert lo == a.min() and hi == a.max()
        a[n // 2] = np.nan
        lo, hi = L.host_minmax(a)
        assert np.isnan(lo) and np.isnan(hi) and np.isnan(a.min())
    lo, hi = L.host_minmax(np.array([np.inf, -np.inf, 3.0]))
    assert lo == -np.inf and hi == np.inf
    lo, hi = L.host_minmax(np.arange(10, dtype=np.int64)[::2])          # any dtype / stride: converted like np.asarray
    assert (lo, hi) == (0.0, 8.0)
    with pytest.raises(ValueError):
        L.host_minmax(np.empty(0))


def test_prefault_leaves_the_buffer_usable():
    """alp_host_prefault is advice to the kernel (huge pages + populate, four threads): whatever the kernel makes of it -- this
    container's refuses nothing, a sandbox may -- the buffer is ordinary memory afterwards; odd addresses and sizes are fine"""
    from alproj_amd import _lib as L
    lib = L.load()
    for nbytes, skew in ((0, 0), (1, 0), (4095, 1), (8192, 0), (3 << 20, 17), ((64 << 20) + 12345, 3)):
        raw = np.empty(nbytes + 64, dtype=np.uint8)
        view = raw[skew:skew + nbytes]
        assert lib.alp_host_prefault(view.ctypes.data_as(L._c_void_p), nbytes, 0) == 0
        assert lib.alp_host_prefault(view.ctypes.data_as(L._c_void_p), nbytes, 7) == 0
        view[:] = 5
        assert int(view.sum()) == 5 * nbytes
    big = L.result_empty((40 << 20,), np.uint8)          # a pool miss: populated before it is handed out
    big[:] = 1
    assert int(big[:: 1 << 20].sum()) == 40


def test_projected_table_is_taken_by_position():
    """rmse / huber_loss read `projected` as the reference does -- projected.to_numpy()[:, 0], [:, 1] (optimize.py:175-176,
    203-206) -- so labels do not matter and an unlabelled frame works (the round-5 advisor's finding: KeyError('u'))"""
    import pandas as pd
    from alproj_amd import optimize as opt
    a = np.arange(12.0).reshape(6, 2)
    for frame in (pd.DataFrame(a), pd.DataFrame(a, columns=["col", "row"]), pd.DataFrame({"v": a[:, 0], "u": a[:, 1]}),
                  pd.DataFrame({"u": a[:, 0], "v": a[:, 1], "extra": 1.0}), pd.DataFrame(a.astype(np.float32))):
        c0, c1 = opt._first_two_columns(frame)
        assert c0.dtype == np.float64 and c0.flags["C_CONTIGUOUS"] and c1.flags["C_CONTIGUOUS"]
        assert np.array_equal(c0, a[:, 0]) and np.array_equal(c1, a[:, 1])
    with pytest.raises(IndexError):
        opt._first_two_columns(pd.DataFrame({"u": a[:, 0]}))
