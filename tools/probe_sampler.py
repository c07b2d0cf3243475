#!/usr/bin/env python3
"""Development probe: ask_population() of the host CMA class with the numpy sampler and with the device
sampler (alp_cma_sample), at the reference's default sigma = 1.0 on [0, 1]^D (almost every draw infeasible)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L            # noqa: E402
from alproj_amd.cma import CMA              # noqa: E402

L.init(0)
for P, D in ((256, 21), (2048, 21), (256, 9), (50, 9)):
    for name, smp in (("numpy ", None), ("device", L.cma_sample)):
        o = CMA(mean=np.full(D, 0.5), sigma=1.0, bounds=np.column_stack([np.zeros(D), np.ones(D)]), population_size=P,
                n_max_resampling=100, seed=1, sampler=smp)
        o.ask_population()
        t = time.perf_counter()
        for _ in range(20):
            X = o.ask_population()
        print(f"pop {P:5d} D {D:2d} sigma 1.0 {name} ask_population: {(time.perf_counter() - t) / 20 * 1e3:8.3f} ms")
