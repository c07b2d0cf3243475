#!/usr/bin/env python3
"""Development probe: device -> host fetch of the projected pixels of an N-vertex float32 set, in its own type and widened
to the reference's float64 (alp_projected_fetch): host-pipelined conversion at several thread counts against the
device conversion.  python3 tools/probe_fetch.py [N] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd import synthetic as syn     # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L.init(0)
n = syn.grid_side(N)
s = syn.surface(n)
xyz = syn.vert_to_xyz_local(s["vert"])
base = syn.local_params(syn.standoff_params(n), s["offsets"])
print(f"host cores: {os.cpu_count()}, result pool cap {L._pool_cap >> 20} MiB")
for prec in ("f32", "f64"):
    pts = L.Points(xyz, [base["x"], base["y"], base["z"]], prec)
    pts.project(L.params_vector(base))
    own, other = (np.float32, np.float64) if prec == "f32" else (np.float64, np.float32)

    def run(tag, dtype, env):
        for k in ("ALP_FETCH_CONVERT", "ALP_HOST_THREADS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            u, v = pts.fetch(dtype)
            ts.append(time.perf_counter() - t)
            del u, v
        gb = 2 * n * n * np.dtype(dtype).itemsize / 1e9
        print(f"  {prec} set -> {np.dtype(dtype).name:8s} {tag:28s} first {ts[0] * 1e3:7.1f} ms, best of the rest {min(ts[1:]) * 1e3:7.1f} ms "
              f"({gb / min(ts[1:]):5.1f} GB/s of result)", flush=True)
        return min(ts[1:])

    plain = run("(no conversion)", own, {})
    for tag, env in (("host, default threads", {}), ("host, 4 threads", {"ALP_HOST_THREADS": "4"}), ("host, 8 threads", {"ALP_HOST_THREADS": "8"}),
                     ("host, 16 threads", {"ALP_HOST_THREADS": "16"}), ("host, 32 threads", {"ALP_HOST_THREADS": "32"}),
                     ("device cast", {"ALP_FETCH_CONVERT": "device"})):
        t = run(tag, other, env)
        print(f"      = {t / plain:.2f} x the plain fetch")
    pts.close()
