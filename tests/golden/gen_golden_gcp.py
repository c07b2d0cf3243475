#!/usr/bin/env python3
"""Golden vectors for set_gcp / filter_gcp_distance FROM THE REFERENCE ITSELF.

Run only in the build container (reference checkout at /root/reference):

    python tests/golden/gen_golden_gcp.py

Loads the reference's ``src/alproj/gcp.py`` by file path (``cv2``, which its module header
imports and neither function touches, is registered as an empty placeholder), runs
``set_gcp`` (gcp.py:614-648) and ``filter_gcp_distance`` (gcp.py:651-726) on seeded synthetic
tables and stores inputs + outputs in ``g10_gcp.npz``.  Only data is written.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference/src/alproj/gcp.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("alproj_ref_gcp", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    gcp = load_reference()
    rng = np.random.default_rng(20260220)
    w, h = 200, 120
    # a reverse_proj table: ~60 % of the pixels see the surface, labels = linear pixel index
    seen = rng.random(w * h) < 0.6
    idx = np.flatnonzero(seen)
    rev = pd.DataFrame({"u": (idx % w).astype("int16"), "v": (idx // w).astype("int16"),
                        "x": 732000.0 + rng.uniform(0, 5000, len(idx)),
                        "y": 4048000.0 + rng.uniform(0, 5000, len(idx)),
                        "z": 1500.0 + rng.uniform(0, 900, len(idx)),
                        "B": rng.integers(0, 255, len(idx)).astype(np.float64),
                        "G": rng.integers(0, 255, len(idx)).astype(np.float64),
                        "R": rng.integers(0, 255, len(idx)).astype(np.float64)}, index=idx)
    # matches: integer pixels, some outside the table, a few outside the image, duplicates
    n = 600
    u_sim = rng.integers(-5, w + 5, n)
    v_sim = rng.integers(-5, h + 5, n)
    u_sim[:20] = u_sim[20:40]
    v_sim[:20] = v_sim[20:40]
    match = pd.DataFrame({"u_org": rng.integers(0, 5616, n), "v_org": rng.integers(0, 3744, n),
                          "u_sim": u_sim, "v_sim": v_sim})
    out = gcp.set_gcp(match, rev)
    # float-typed matches (what a sub-pixel matcher returns after rounding) incl. non-integral ones
    match_f = match.astype(np.float64)
    match_f.loc[match_f.index[::7], "u_sim"] += 0.5
    out_f = gcp.set_gcp(match_f, rev)

    cam = {"x": 732100.0, "y": 4050500.0, "z": 2400.0}
    g = out.copy()
    g.iloc[5, g.columns.get_loc("x")] = np.nan
    g.iloc[9, g.columns.get_loc("z")] = np.nan
    cases = [(None, None), (1500.0, None), (None, 3000.0), (1200.0, 2800.0), (0.0, 1e9)]
    res = {}
    for k, (lo, hi) in enumerate(cases):
        f = gcp.filter_gcp_distance(g, cam, min_distance=lo, max_distance=hi)
        res[f"filt{k}_values"] = f.to_numpy(dtype=np.float64)
        res[f"filt{k}_index"] = f.index.to_numpy()
    np.savez_compressed(
        os.path.join(OUT, "g10_gcp.npz"),
        w=w, h=h, rev_index=idx, rev_xyz=rev[["x", "y", "z"]].to_numpy(),
        match=match.to_numpy(), match_f=match_f.to_numpy(),
        set_values=out.to_numpy(dtype=np.float64), set_index=out.index.to_numpy(),
        set_columns=np.array(list(out.columns)),
        setf_values=out_f.to_numpy(dtype=np.float64), setf_index=out_f.index.to_numpy(),
        filt_input=g.to_numpy(dtype=np.float64), filt_input_index=g.index.to_numpy(),
        cam=np.array([cam["x"], cam["y"], cam["z"]]),
        filt_cases=np.array([[np.nan if a is None else a, np.nan if b is None else b] for a, b in cases]),
        **res)
    print("g10_gcp.npz:", len(out), "of", n, "matches kept;", {k: v.shape for k, v in res.items() if "values" in k})


if __name__ == "__main__":
    main()
