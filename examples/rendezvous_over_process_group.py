"""Example: hand the 128-byte RCCL unique id of libalproj_hip.so over a process group that a foreign launcher
(``python -m torch.distributed.run``) has already initialised, instead of the launcher of ``alproj_amd.launch``.

Not part of the product (``alproj_amd`` imports no deep-learning framework): the product's contract is
``alproj_amd.dist.init_comm(rank, world, bcast_bytes, device)``, where ``bcast_bytes(b) -> bytes`` returns rank 0's
argument on every rank.  Any control plane can supply that callable; this file supplies it from ``torch.distributed``.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 your_script.py
    # in your_script.py:
    #   import torch.distributed as dist; dist.init_process_group("gloo")
    #   from rendezvous_over_process_group import init_from_process_group
    #   rank, world = init_from_process_group()
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def init_from_process_group(device=None):
    """Use an already initialised torch.distributed process group as the control plane of ``init_comm``."""
    import torch
    import torch.distributed as dist
    from alproj_amd import dist as adist

    rank, world = dist.get_rank(), dist.get_world_size()

    def bcast(b):
        t = torch.tensor(list(b), dtype=torch.uint8)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())

    adist.init_comm(rank, world, bcast, device)
    return rank, world
