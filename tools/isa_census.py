#!/usr/bin/env python3
"""Instruction census of the hot loop of popeval_kernel<T, HUBER> (K2): compiles alp_points.hip to
gfx950 assembly and counts the instructions of the innermost candidate loop (the full group of V points).
   python3 tools/isa_census.py > profiles/r02_popeval_isa_census.txt            # float, V = 6
   python3 tools/isa_census.py f64 [-DPOP_VD=3 ...] > profiles/r05_popeval_f64_isa_census.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

F64 = len(sys.argv) > 1 and sys.argv[1] in ("f64", "lf64")
LF = len(sys.argv) > 1 and sys.argv[1] in ("lf", "lf64")          # the lens-free variant (round 6)
DEFS = [a for a in sys.argv[1:] if a.startswith("-D")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(tempfile.mkdtemp(), "points.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/alproj_amd/csrc",
                "--cuda-device-only", "-S", f"{ROOT}/alproj_amd/csrc/alp_points.hip", "-o", asm] + DEFS, check=True,
               stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
sym = r"^_ZN3alp14popeval_kernelIdLi1ENS_6PopCfgIdEELb0EdLb0EE.*:" if F64 else r"^_ZN3alp14popeval_kernelIfLi1ENS_6PopCfgIfEELb0EfLb0EE.*:"
if LF:
    sym = r"^_ZN3alp14popeval_kernelIdLi1ENS_8PopCfgLFIdEELb0EdLb1EE.*:" if F64 else r"^_ZN3alp14popeval_kernelIfLi1ENS_8PopCfgLFIfEELb0EfLb1EE.*:"
start = next(i for i, l in enumerate(lines) if re.match(sym, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# innermost loops = Depth=3 loop headers; the first one is the full group of V points
heads = [i for i, l in enumerate(body) if "Depth=3" in l and l.startswith(".LBB")]
first = heads[0]
last = next(i for i in range(first + 1, len(body)) if "s_cbranch" in body[i] and i > first + 50)
loop = [l.split()[0] for l in body[first + 1:last + 1] if l.strip() and not l.strip().startswith((";", "."))]
hist = collections.Counter(loop)
valu = sum(v for k, v in hist.items() if k.startswith("v_"))
V = 8 if LF else 6
if F64:
    V = 6 if LF else 5
    for d in DEFS:
        if d.startswith("-DPOP_VD="):
            V = int(d.split("=")[1])
print(f"popeval_kernel<{'double' if F64 else 'float'}, HUBER, V = {V}, TC = 128{', LENS_FREE' if LF else ''}>{' ' + ' '.join(DEFS) if DEFS else ''}: "
      f"innermost loop = ONE candidate against the {V} points of a lane")
print(f"instructions in the loop body: {len(loop)}; vector ALU: {valu} = {valu / V:.1f} per evaluation\n")
groups = collections.OrderedDict([
    ("fused multiply-add (v_fma_f32, v_fmac_f32)", ("v_fma_f32", "v_fmac_f32")),
    ("multiply / add / min (full rate)", ("v_mul_f32", "v_add_f32_e", "v_min_f32", "v_sub_f32", "v_max_f32")),
    ("quarter-rate transcendentals (v_rcp_f32, v_sqrt_f32)", ("v_rcp_f32", "v_sqrt_f32")),
    ("float64 fused multiply-add (v_fma_f64, v_fmac_f64)", ("v_fma_f64", "v_fmac_f64")),
    ("float64 multiply / add / min / max", ("v_mul_f64", "v_add_f64", "v_min_f64", "v_max_f64")),
    ("float64 transcendentals (v_rcp_f64, v_rsq_f64) and fix-ups (v_div_fixup_f64)", ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_fixup_f64")),
    ("float64 compares (v_cmp_*_f64, v_cmp_class_f64)", ("v_cmp",)),
    ("cross-lane reduction of the candidate's sum (DPP adds / moves, ds_bpermute)", ("v_add_f32_dpp", "v_mov_b32_dpp", "ds_bpermute", "v_add_f64_dpp")),
    ("float64 accumulation (v_cvt_f64_f32)", ("v_cvt_f64",)),
    ("register moves / selects (v_mov_b32, v_cndmask)", ("v_mov_b32_e", "v_cndmask")),
    ("LDS: pose record reads (ds_read_b128 / b64), accumulator read-modify-write", ("ds_",)),
    ("waits and hazards (s_waitcnt, s_nop)", ("s_waitcnt", "s_nop")),
    ("scalar / control", ("s_",)),
])
seen = set()
for name, prefixes in groups.items():
    n = 0
    for k, v in hist.items():
        if k in seen:
            continue
        if any(k.startswith(p) for p in prefixes):
            n += v
            seen.add(k)
    print(f"  {n:4d}  ({n / V:5.2f} per evaluation)  {name}")
rest = {k: v for k, v in hist.items() if k not in seen}
print(f"  other: {rest}")
print("\nraw histogram:")
for k, v in hist.most_common():
    print(f"  {v:4d}  {k}")
