"""alproj_amd -- MI355X (gfx950) implementation of the alproj camera-projection hot path.

Sub-modules mirror the reference package layout for the functions on the path:

* ``alproj_amd.optimize`` -- project, rmse, huber_loss, compute_residuals, bounds_to_array,
  CMAOptimizer, LsqOptimizer               (reference: src/alproj/optimize.py)
* ``alproj_amd.project``  -- projection_mat, modelview_mat, distort, persp_proj, sim_image,
  reverse_proj                             (reference: src/alproj/project.py)
* ``alproj_amd.cma``      -- the CMA-ES sampler the reference takes from the ``cmaes`` package
* ``alproj_amd.dist``     -- one-process-per-GPU sharding of the vertex array + RCCL setup
* ``alproj_amd.launch``   -- torch-free launcher and localhost control plane of a multi-GPU job (what ``bench.py --gpus N`` uses)
* ``alproj_amd.synthetic``-- the synthetic DSM / camera used by tests and bench.py

All per-point work goes through ``libalproj_hip.so`` (ctypes; see include/alproj_hip.h);
there is no CPU fallback.
"""
__version__ = "0.4.0"

from .gcp import filter_gcp_distance      # noqa: E402,F401  (the reference's package re-exports it: src/alproj/__init__.py:1)
