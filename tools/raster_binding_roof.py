#!/usr/bin/env python3
"""What binds the render frame when it is not HBM: the per-frame floor set by the two resources its kernels saturate.

    python3 tools/raster_binding_roof.py profiles/rNN_raster_frame_kernels_sq.csv profiles/rNN_raster_frame_kernels_atomics.csv > profiles/rNN_raster_binding_roof.json

From the per-kernel counter tables of one probe (`tools/profile_round.sh`, `rocprofv3 --pmc` in separate passes; the probe draws
`frames` frames = the dispatch count of `resolve_kernel`):

* atomic floor  = 64-byte atomic line-requests at L2 per frame (TCC_ATOMIC_sum over the frame's kernels) / ATOMIC_RATE, the
  22-27 G requests/s wall measured with tools/atomic_rate.hip on the same box class (profiles/r04_parked_tiles_lab.txt; 24.5 G/s
  is its centre),
* vector floor  = SQ_ACTIVE_INST_VALU (quad-cycles: x 4 cycles) summed over the frame's kernels / (1024 SIMDs x 2.4 GHz): the
  time the vector pipes would need if that work were spread evenly over all of them with nothing else in the way.

bench.py divides max(atomic floor, vector floor) by the frame time it measures: `raster_binding_roof_frac`."""
import csv
import json
import sys

ATOMIC_RATE = 24.5e9      # 64-byte atomic line-requests / s at L2 (measured wall: 22-27 G/s)
SIMDS = 256 * 4
CLOCK_HZ = 2.4e9


def table(path):
    """the second table of tools/rocpd_summary.py --csv: kernel, counter, mean_per_dispatch, dispatches"""
    rows, on = [], False
    for r in csv.reader(open(path)):
        if r[:2] == ["kernel", "counter"]:
            on = True
        elif on and len(r) == 4:
            rows.append((r[0], r[1], float(r[2]), int(r[3])))
    return rows


def per_frame(rows, counter, skip=("tile_bounds_kernel",)):       # tile_bounds runs once per mesh, not per frame
    frames = max(d for k, c, _, d in rows if "resolve_kernel" in k)
    per_kernel = {k: m * d / frames for k, c, m, d in rows if c == counter and not any(s in k for s in skip)}
    return frames, per_kernel


def main():
    sq, at = sys.argv[1], sys.argv[2]
    frames, valu = per_frame(table(sq), "SQ_ACTIVE_INST_VALU")
    _, atom = per_frame(table(at), "TCC_ATOMIC_sum")
    requests = sum(atom.values())
    quad = sum(valu.values())
    out = {
        "frames_in_probe": frames,
        "atomic_line_requests_per_frame": requests, "atomic_rate_per_s": ATOMIC_RATE,
        "atomic_floor_ms": requests / ATOMIC_RATE * 1e3,
        "valu_active_quad_cycles_per_frame": quad, "simds": SIMDS, "clock_hz": CLOCK_HZ,
        "valu_floor_ms": quad * 4 / (SIMDS * CLOCK_HZ) * 1e3,
        "by_kernel": {k: {"atomic_line_requests": atom.get(k, 0.0), "valu_active_quad_cycles": valu.get(k, 0.0)}
                      for k in sorted(set(valu) | set(atom))},
        "sources": [sq, at],
        "note": "floor_ms = max(atomic_floor_ms, valu_floor_ms); bench.py reports floor_ms / measured frame time as raster_binding_roof_frac",
    }
    out["binding"] = "l2_atomics" if out["atomic_floor_ms"] >= out["valu_floor_ms"] else "valu"
    out["floor_ms"] = max(out["atomic_floor_ms"], out["valu_floor_ms"])
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
