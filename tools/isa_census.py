#!/usr/bin/env python3
"""Instruction census of the hot loop of popeval_kernel<float, HUBER> (K2): compiles alp_points.hip to
gfx950 assembly and counts the instructions of the innermost candidate loop (the V = 6 group).
   python3 tools/isa_census.py > profiles/r02_popeval_isa_census.txt"""
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(tempfile.mkdtemp(), "points.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/alproj_amd/csrc",
                "--cuda-device-only", "-S", f"{ROOT}/alproj_amd/csrc/alp_points.hip", "-o", asm], check=True,
               stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3alp14popeval_kernelIfLi1ENS_6PopCfgIfEELb0EfEE.*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# innermost loops = Depth=3 loop headers; the first one is the full group of V points
heads = [i for i, l in enumerate(body) if "Depth=3" in l and l.startswith(".LBB")]
first = heads[0]
last = next(i for i in range(first + 1, len(body)) if "s_cbranch" in body[i] and i > first + 50)
loop = [l.split()[0] for l in body[first + 1:last + 1] if l.strip() and not l.strip().startswith((";", "."))]
hist = collections.Counter(loop)
valu = sum(v for k, v in hist.items() if k.startswith("v_"))
V = 6
print("popeval_kernel<float, HUBER, V = 6, TC = 128>: innermost loop = ONE candidate against the 6 points of a lane")
print(f"instructions in the loop body: {len(loop)}; vector ALU: {valu} = {valu / V:.1f} per evaluation\n")
groups = collections.OrderedDict([
    ("fused multiply-add (v_fma_f32, v_fmac_f32)", ("v_fma_f32", "v_fmac_f32")),
    ("multiply / add / min (full rate)", ("v_mul_f32", "v_add_f32_e", "v_min_f32", "v_sub_f32", "v_max_f32")),
    ("quarter-rate transcendentals (v_rcp_f32, v_sqrt_f32)", ("v_rcp_f32", "v_sqrt_f32")),
    ("cross-lane reduction of the candidate's sum (DPP adds / moves)", ("v_add_f32_dpp", "v_mov_b32_dpp")),
    ("float64 accumulation (v_cvt_f64_f32, v_add_f64)", ("v_cvt_f64", "v_add_f64")),
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
