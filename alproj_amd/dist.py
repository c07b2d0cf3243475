"""One process per GPU: sharding of the vertex array and set-up of the RCCL communicator.

The only path that shards is the population evaluation: every rank holds a
contiguous range of the DSM vertices (+ their observed pixels) resident in its GPU, evaluates
ALL candidates on its range, and the per-candidate partial sums plus the vertex counts are
summed with ONE ``ncclAllReduce(sum, float64, P + 1)`` per generation inside
``alp_eval_population`` (libalproj_hip.so, on the library stream).  The forward projection
shards the same way with no collective at all; the CMA-ES sampler is replicated (same seed,
same losses on every rank -> same trajectory).

The 128-byte RCCL unique id has to travel from rank 0 to the others once; this module takes
it over whatever control plane the launcher provides:

* ``init_comm(rank, world, bcast_bytes)`` -- any callable that hands rank 0's bytes to every rank
  (``alproj_amd.launch.Control.bcast_bytes`` is what ``bench.py`` passes; ``examples/rendezvous_over_process_group.py``
  shows the same over a process group a foreign launcher already made),
* ``init_from_file(path)`` -- a file on a shared filesystem,

The reduction contract itself (``combine_partials``) is plain numpy so that it can be tested
on CPU with the gloo backend (tests/test_dist_gloo.py).
"""
import os
import sys
import time

import numpy as np

from . import _lib


def shard_bounds(n, rank, world_size):
    """Contiguous, near-equal split of n items: rank r gets [lo, hi)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return n * rank // world_size, n * (rank + 1) // world_size


def shard_rows(n_rows, rank, world_size):
    """Row-wise split of a grid DSM (whole rows per rank, so shards stay rectangular)."""
    return shard_bounds(n_rows, rank, world_size)


def pack_partials(loss_sums, n_local):
    """The P + 1 doubles a rank contributes: per-candidate SUMS (not means) + its vertex count."""
    out = np.empty(len(loss_sums) + 1, dtype=np.float64)
    out[:-1] = loss_sums
    out[-1] = n_local
    return out


def combine_partials(reduced):
    """From the all-reduced vector: (mean loss per candidate, argmin).

    Mean = sum / total count, exactly ``np.mean`` over the union of the shards; a NaN partial
    poisons that candidate on every rank (quirk Q7); argmin = first index, NaN never wins unless
    every loss is NaN (then 0) -- the same rule as ``alp_eval_population_wait``."""
    reduced = np.asarray(reduced, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        losses = reduced[:-1] / reduced[-1]
    finite = ~np.isnan(losses)
    amin = int(np.flatnonzero(finite)[np.argmin(losses[finite])]) if finite.any() else 0
    return losses, amin


# ------------------------------------------------------------------------------------------
# RCCL communicator of libalproj_hip.so
# ------------------------------------------------------------------------------------------
def init_comm(rank, world_size, bcast_bytes, device=None):
    """Initialise the library on ``device`` (default LOCAL_RANK) and create the communicator.
    ``bcast_bytes(b: bytes) -> bytes`` must return rank 0's argument on every rank.

    A launcher may give every rank a device of its own by restricting what the rank SEES (``HIP_VISIBLE_DEVICES=k`` per
    rank): then the one visible device is number 0 whatever LOCAL_RANK says."""
    if world_size > 1:
        want = device if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
        if want >= 1 and _lib.device_count() == 1:
            print(f"alproj_amd.dist: rank {rank} sees ONE device ({os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES'))!r} "
                  f"visible): using it as device 0 instead of device {want}", file=sys.stderr, flush=True)
            device = 0
    _lib.init(device)
    if world_size <= 1:
        return
    uid = _lib.comm_unique_id() if rank == 0 else b"\0" * _lib.UNIQUE_ID_BYTES
    uid = bcast_bytes(uid)
    _lib.comm_init(uid, rank, world_size)


def init_from_file(path, rank, world_size, device=None, timeout_s=120.0):
    """Rendezvous through ``path`` on a filesystem all ranks see (rank 0 writes, others poll)."""
    def bcast(b):
        if rank == 0:
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(b)
            os.replace(tmp, path)
            return b
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == _lib.UNIQUE_ID_BYTES:
                    return data
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"no RCCL unique id at {path} after {timeout_s}s")
            time.sleep(0.05)

    init_comm(rank, world_size, bcast, device)
    return rank, world_size


def shutdown():
    _lib.comm_destroy()
