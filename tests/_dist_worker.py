"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo group on CPU.

Each rank takes its contiguous shard of the points (alproj_amd.dist.shard_bounds), computes
the per-candidate loss SUMS of its shard (with the CPU oracle standing in for the kernel: this
test is about the sharding/reduction logic, not the arithmetic), contributes P + 1 doubles to
one all-reduce, and derives mean losses + argmin exactly like the GPU path does.
usage: _dist_worker.py RANK WORLD PORT OUT_NPZ
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
    import torch
    import torch.distributed as dist
    from alproj_amd import dist as adist
    from alproj_amd import synthetic as syn
    from oracle import ref_numpy as orc

    dist.init_process_group("gloo", rank=rank, world_size=world)
    truth = syn.truth_params(316)
    init = syn.base_params(316)
    n = 3001                                  # not divisible by the world size
    xyz = syn.gcp_points(n, truth, seed=5)
    uv = orc.project_points(xyz, truth) + np.random.default_rng(5).normal(0, 1.0, (n, 2))
    xyz[17] = [init["x"], init["y"], init["z"]]       # AT candidate 0's camera: NaN partial (Q7)
    bounds = orc.bounds_to_array(init, syn.TARGETS_D9)
    X = np.random.default_rng(9).uniform(0.4, 0.6, (12, 9))
    X[0] = 0.5                                        # candidate 0 == init pose
    lo, hi = adist.shard_bounds(n, rank, world)
    sums = np.empty(len(X))
    for i, x in enumerate(X):
        p = orc.candidate_params(init, syn.TARGETS_D9, bounds, x)
        with np.errstate(all="ignore"):
            sums[i] = orc.huber(uv[lo:hi], orc.project_points(xyz[lo:hi], p), 10.0) * (hi - lo)
    t = torch.from_numpy(adist.pack_partials(sums, hi - lo))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)          # the ONE collective of a generation
    losses, amin = adist.combine_partials(t.numpy())
    np.savez(out, losses=losses, amin=amin, lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
