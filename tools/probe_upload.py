"""Upload rate of alp_points_create / alp_mesh_create at 100 M points (dev probe)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from alproj_amd import _lib as L
L.init(0)
n = 100_000_000
xyz = np.random.default_rng(0).random((n, 3), dtype=np.float32)
for rep in range(4):
    t = time.perf_counter()
    p = L.Points(xyz, [0.0, 0.0, 0.0], "f32")
    dt = time.perf_counter() - t
    print(f"Points f32<-f32 create {dt*1e3:.1f} ms  {xyz.nbytes/dt/1e9:.1f} GB/s")
    p.close()
for rep in range(3):
    t = time.perf_counter()
    m = L.Mesh(xyz, None, None, grid=(10000, 10000))
    dt = time.perf_counter() - t
    print(f"Mesh create {dt*1e3:.1f} ms  {xyz.nbytes/dt/1e9:.1f} GB/s")
    m.close()
x64 = xyz[: n // 2].astype(np.float64)
for rep in range(3):
    t = time.perf_counter()
    p = L.Points(x64, [0.0, 0.0, 0.0], "f64")
    dt = time.perf_counter() - t
    print(f"Points f64<-f64 50M create {dt*1e3:.1f} ms  {x64.nbytes/dt/1e9:.1f} GB/s")
    p.close()
