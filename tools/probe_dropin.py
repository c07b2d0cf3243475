"""Development: where the first call of the reference-typed pair spends its time (100 M vertices: float64 vert / col,
int64 ind = 9.6 GB of host arrays).   python3 tools/probe_dropin.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alproj_amd import _lib as L
from alproj_amd import project as aproj
from alproj_amd import synthetic as syn
L.init(0)
n = 10000
s = syn.surface(n)
vert = s["vert"].astype(np.float64)
col = np.random.default_rng(1).random((n * n, 3))
ind = syn.grid_indices(n, np.int64)
cam = syn.base_params(n)
print("arrays ready", flush=True)

def t(label, fn, nbytes=None):
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print(f"{label:50s} {dt * 1e3:8.1f} ms" + (f"  {nbytes / dt / 1e9:6.1f} GB/s" if nbytes else ""), flush=True)
    return r

for rep in range(2):
    print(f"--- rep {rep}")
    t("digest vert", lambda: L.host_hash64(vert), vert.nbytes)
    t("digest ind", lambda: L.host_hash64(ind), ind.nbytes)
    m = t("Mesh(vert f64, None, grid)", lambda: L.Mesh(vert, None, None, grid=(n, n)), vert.nbytes); m.close()
    m = t("Mesh(vert f64, col f64, grid)", lambda: L.Mesh(vert, col, None, grid=(n, n)), vert.nbytes + col.nbytes); m.close()
    m = t("Mesh(vert f64, None, ind int64)", lambda: L.Mesh(vert, None, ind), vert.nbytes + ind.nbytes); m.close()
    m = t("Mesh(vert f64, col f64, ind int64)", lambda: L.Mesh(vert, col, ind), vert.nbytes + col.nbytes + ind.nbytes)
    t("first frame", lambda: (m.render_enqueue(L.params_vector(cam), s["offsets"]), L.synchronize()))
    t("fetch_u8", lambda: m.fetch_u8())
    t("close", m.close)
    aproj.clear_mesh_cache()
    sim = t("sim_image (whole first call)", lambda: aproj.sim_image(vert, col, ind, cam, s["offsets"]), vert.nbytes + col.nbytes + ind.nbytes)
    print("   ", {k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in aproj.LAST_TIMING.items()})
    t("reverse_proj (second call)", lambda: aproj.reverse_proj(sim, vert, ind, cam, s["offsets"]))
    print("   ", {k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in aproj.LAST_TIMING.items()})
    aproj.clear_mesh_cache()

print("--- pieces of _resident_mesh")
for rep in range(2):
    aproj.clear_mesh_cache()
    m = t("  Mesh(vert, col, ind)", lambda: L.Mesh(vert, col, ind), vert.nbytes + col.nbytes + ind.nbytes)
    t("  _key(vert)", lambda: aproj._key(vert))
    t("  _key(ind)", lambda: aproj._key(ind))
    t("  _key(col)", lambda: aproj._key(col))
    t("  frame_counts", m.frame_counts)
    m.close()
    t("  _resident_mesh", lambda: aproj._resident_mesh(vert, col, ind, None))
    aproj.clear_mesh_cache()
