#!/usr/bin/env python3
"""One-command proof that tests/golden/*.npz are what the REFERENCE produces (build container only).

    python tools/check_golden_reproducible.py [--skip-c4] [--keep]

Copies the generators and the seeded scene builders they import (tests/ without its fixtures, alproj_amd/*.py) into a
temporary tree, runs every generator there -- each imports /root/reference/src/alproj/*.py by file path and writes its
.npz next to itself, i.e. into the temporary tree -- and compares every array of every regenerated file with the
committed fixture: same names, same dtype, same shape, same bytes (NaNs compare equal to NaNs of the same bit pattern
because the comparison is on the raw bytes).  One line per file; exit status 1 on any difference, 2 when the reference
checkout is absent.  `--skip-c4` leaves out g16 (the 5616 x 3744 frame through llvmpipe: ~40 s and ~10 GB).

Nothing of the product or the test suite imports this file; it reads /root/reference, so it cannot run on the GPU box.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference/src/alproj"

# generator -> the fixtures it writes
GENERATORS = [
    ("gen_golden.py", ["g1_matrices", "g2_distort", "g3_project", "g4_losses", "g5_population", "g6_bounds",
                       "g7_gl_matrices", "g8_residuals"]),
    ("gen_golden_geotiff.py", ["g9_geotiff"]),
    ("gen_golden_gcp.py", ["g10_gcp"]),
    ("gen_golden_surface.py", ["g11_surface"]),
    ("gen_golden_render.py", ["g12_wrappers", "g13_distort_map", "g14_lsq"]),
    ("gen_golden_gl.py", ["g15_gl_render"]),
    ("gen_golden_gl_c4.py", ["g16_gl_c4_frame"]),
    ("gen_golden_geotiff_float.py", ["g17_geotiff_float"]),
    ("gen_golden_geotiff_bytes.py", ["g18_geotiff_bytes"]),
    ("gen_golden_first_phase.py", ["g19_first_phase"]),
]


def compare(committed, regenerated):
    """-> (arrays compared, list of differences)"""
    a, b = np.load(committed, allow_pickle=False), np.load(regenerated, allow_pickle=False)
    diffs = []
    if sorted(a.files) != sorted(b.files):
        diffs.append(f"array names differ: only committed {sorted(set(a.files) - set(b.files))}, "
                     f"only regenerated {sorted(set(b.files) - set(a.files))}")
    n = 0
    for k in sorted(set(a.files) & set(b.files)):
        x, y = a[k], b[k]
        n += 1
        if x.dtype != y.dtype or x.shape != y.shape:
            diffs.append(f"{k}: {x.dtype}{x.shape} committed vs {y.dtype}{y.shape} regenerated")
        elif np.ascontiguousarray(x).tobytes() != np.ascontiguousarray(y).tobytes():
            bad = int(np.sum(np.ascontiguousarray(x).view(np.uint8) != np.ascontiguousarray(y).view(np.uint8))) if x.dtype.kind != "U" else -1
            diffs.append(f"{k}: {bad} bytes differ")
    return n, diffs


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--skip-c4", action="store_true", help="leave out g16 (config 4's frame through llvmpipe)")
    ap.add_argument("--keep", action="store_true", help="keep the temporary tree")
    args = ap.parse_args()
    if not os.path.isdir(REFERENCE):
        print(f"check_golden_reproducible: {REFERENCE} is absent -- this runs in the build container only", file=sys.stderr)
        return 2
    tmp = tempfile.mkdtemp(prefix="alproj_golden_")
    try:
        shutil.copytree(os.path.join(ROOT, "tests"), os.path.join(tmp, "tests"),
                        ignore=shutil.ignore_patterns("*.npz", "__pycache__", "*.so"))
        shutil.copytree(os.path.join(ROOT, "alproj_amd"), os.path.join(tmp, "alproj_amd"),
                        ignore=shutil.ignore_patterns("__pycache__", "*.so", "csrc"))
        gdir = os.path.join(tmp, "tests", "golden")
        assert not [f for f in os.listdir(gdir) if f.endswith(".npz")]
        files = arrays = bad = 0
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for gen, fixtures in GENERATORS:
            if args.skip_c4 and gen == "gen_golden_gl_c4.py":
                print(f"SKIPPED   {fixtures[0]}.npz ({gen}: --skip-c4)")
                continue
            t = time.time()
            r = subprocess.run([sys.executable, os.path.join(gdir, gen)], cwd=tmp, env=env, capture_output=True, text=True)
            dt = time.time() - t
            if r.returncode != 0:
                print(f"FAILED    {gen}: exit {r.returncode}\n{r.stderr[-2000:]}")
                bad += len(fixtures)
                continue
            for fx in fixtures:
                new, old = os.path.join(gdir, fx + ".npz"), os.path.join(GOLDEN, fx + ".npz")
                if not os.path.exists(new) or not os.path.exists(old):
                    print(f"MISSING   {fx}.npz ({'not regenerated' if not os.path.exists(new) else 'not committed'})")
                    bad += 1
                    continue
                n, diffs = compare(old, new)
                files += 1
                arrays += n
                if diffs:
                    bad += 1
                    print(f"DIFFERENT {fx}.npz ({gen}, {dt:.0f} s): " + "; ".join(diffs[:6]))
                else:
                    print(f"identical {fx}.npz: {n} arrays ({gen}, {dt:.0f} s)")
        extra = sorted(set(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz")) - {fx for _, fs in GENERATORS for fx in fs})
        for fx in extra:
            print(f"NO GENERATOR for committed fixture {fx}.npz")
            bad += 1
        print(f"{files} files, {arrays} arrays compared with the committed fixtures: " + ("ALL IDENTICAL" if not bad else f"{bad} PROBLEMS"))
        return 1 if bad else 0
    finally:
        if args.keep:
            print("temporary tree kept:", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
