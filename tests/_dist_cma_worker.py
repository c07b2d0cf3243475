"""Worker of tests/test_dist_gloo.py::test_optimisers_*: one rank of a world_size-N gloo group on CPU running the
PRODUCT's optimiser loops (alproj_amd.optimize.CMAOptimizer.optimize / LsqOptimizer.optimize, including their
multi-rank branches: seed broadcast, per-generation candidate broadcast, final collective) over its shard of the
points.  Only the device is stood in for: the communicator calls of alproj_amd._lib go to gloo, and the point set
is a CPU object whose evaluation is the oracle on the shard + ONE all-reduce of P + 1 doubles, packed and combined
like the library does (alproj_amd.dist).  Test infrastructure: the product has no such path.
usage: _dist_cma_worker.py RANK WORLD PORT OUT_NPZ
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import pandas as pd
    import torch
    import torch.distributed as dist
    from alproj_amd import _lib
    from alproj_amd import dist as adist
    from alproj_amd import optimize as aopt
    from alproj_amd import synthetic as syn
    from oracle import ref_numpy as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []

    def comm_bcast(array, root=0):
        t = torch.from_numpy(array.reshape(-1).view(np.uint8))     # bytes, in place, like alp_comm_bcast
        dist.broadcast(t, src=root)
        log.append(("bcast", str(array.dtype), tuple(array.shape)))
        return array

    def comm_allgather(array):
        """rows of every rank in rank order, like alp_comm_allgather_counts + alp_comm_allgatherv"""
        array = np.ascontiguousarray(array)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([array.shape[0]], dtype=torch.int64))
        counts = [int(c[0]) for c in counts]
        width = array.shape[1:] if array.ndim > 1 else ()
        pad = np.zeros((max(counts),) + width, dtype=array.dtype)
        pad[:array.shape[0]] = array
        parts = [torch.zeros(pad.shape, dtype=torch.from_numpy(pad).dtype) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(pad))
        log.append(("allgather", tuple(array.shape)))
        return np.concatenate([p.numpy()[:c] for p, c in zip(parts, counts)])

    _lib.comm_info = lambda: (rank, world)
    _lib.comm_bcast = comm_bcast
    _lib.comm_allgather = comm_allgather

    truth = syn.truth_params(316)
    init = dict(truth, pan=truth["pan"] + 1.5, tilt=truth["tilt"] - 1.0, fov=truth["fov"] + 2, x=truth["x"] + 3)
    n = 1201
    xyz = syn.gcp_points(n, truth, seed=11)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(11).normal(0, 0.8, (n, 2))
    lo, hi = adist.shard_bounds(n, rank, world)

    class ShardPoints:
        """what alproj_amd._lib.Points is to the optimisers, on this rank's shard"""
        precision, n = _lib.ALP_F64, hi - lo

        def eval_population(self, cand, kind, f_scale, want_argmin=True):
            sums = np.empty(len(cand))
            for i, c in enumerate(cand):
                p = orc.vector_to_params(c)
                proj = orc.project_points(xyz[lo:hi], p)
                loss = orc.mean_distance(uv[lo:hi], proj) if kind == _lib.LOSS_MEAN_DIST else orc.huber(uv[lo:hi], proj, f_scale)
                sums[i] = loss * (hi - lo)
            t = torch.from_numpy(adist.pack_partials(sums, hi - lo))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            log.append(("eval", len(cand)))
            return adist.combine_partials(t.numpy())

        def residuals(self, vec):
            return orc.residual_vector(xyz[lo:hi], uv[lo:hi], orc.vector_to_params(vec))

        def residuals_batch(self, cand):
            return np.stack([self.residuals(c) for c in cand])

        def close(self):
            pass

    aopt.BaseOptimizer._device_points = lambda self, precision: ShardPoints()
    obj = pd.DataFrame(xyz, columns=["x", "y", "z"])
    img = pd.DataFrame(uv, columns=["u", "v"])
    res = {"init_err": orc.mean_distance(uv, orc.project_points(xyz, init))}
    # ---- CMA-ES, nobody passes a seed (the reference's call): rank 0's entropy must reach every rank
    o = aopt.CMAOptimizer(obj, img, init)
    o.set_target(syn.TARGETS_D9)
    X_seen = []
    ask = aopt.CMA.ask_population
    tell = aopt.CMA.tell_population

    def tell_spy(self, X, losses):
        X_seen.append(np.array(X, copy=True))
        return tell(self, X, losses)

    aopt.CMA.tell_population = tell_spy
    params, err = o.optimize(generation=20, sigma=0.3, population_size=12, f_scale=10.0, seed=None, progress=False)
    aopt.CMA.tell_population = tell
    res["cma_params"] = np.array([params[k] for k in _lib.PARAM_KEYS], dtype=np.float64)
    res["cma_err"] = err
    res["cma_X"] = np.stack(X_seen)
    res["cma_log"] = np.array([repr(e) for e in log])
    # ---- least squares: every rank gathers all shards' residuals / Jacobian rows and solves the reference's ONE problem
    for tag, kw in (("lsq", dict(method="trf", loss="linear", max_nfev=30)),
                    ("lsq_huber", dict(method="trf", loss="huber", f_scale=2.0, max_nfev=30)),
                    ("lsq_2point", dict(method="dogbox", loss="linear", jac="2-point", max_nfev=30))):
        del log[:]
        q = aopt.LsqOptimizer(obj, img, init)
        q.set_target(["fov", "pan", "tilt", "roll"])
        lp, lerr = q.optimize(**kw)
        res[tag + "_params"] = np.array([lp[k] for k in _lib.PARAM_KEYS], dtype=np.float64)
        res[tag + "_err"] = lerr
        res[tag + "_log"] = np.array([repr(e) for e in log])
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
