#!/usr/bin/env python3
"""Development: merge the line lists written by tests/conftest.py under ALP_PY_COVERAGE and report, per file of alproj_amd/,
the lines of function bodies (from the compiled code objects) no session executed.  Lines that only run in child
processes (launch.py's ranks, the gloo workers) are not seen.
   ALP_PY_COVERAGE=/tmp/cpu.txt python3 -m pytest tests -m "not gpu" -q;  ALP_PY_COVERAGE=gpurun_out/gpu.txt python3 -m pytest tests -m gpu -q
   python3 tools/py_coverage.py /tmp/cpu.txt gpurun_out/gpu.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seen = set()
for f in sys.argv[1:]:
    for line in open(f):
        p, _, n = line.strip().rpartition(":")
        seen.add((p, int(n)))


def executable_lines(path):
    lines = set()
    todo = [compile(open(path).read(), path, "exec")]
    while todo:
        co = todo.pop()
        if co.co_flags & 0x1:            # CO_OPTIMIZED: a function body.  Module and class bodies run at import, before the collector starts
            lines.update(ln for _, _, ln in co.co_lines() if ln and ln != co.co_firstlineno)
        todo.extend(c for c in co.co_consts if hasattr(c, "co_lines"))
    return lines


tot = hit = 0
for name in sorted(os.listdir(os.path.join(ROOT, "alproj_amd"))):
    if not name.endswith(".py"):
        continue
    rel = os.path.join("alproj_amd", name)
    ex = executable_lines(os.path.join(ROOT, rel))
    got = {n for p, n in seen if p == rel}
    src = open(os.path.join(ROOT, rel)).read().split("\n")
    # a def / class line and a docstring line count as executed when the module was imported: only body lines matter
    miss = sorted(n for n in ex - got)
    tot += len(ex)
    hit += len(ex) - len(miss)
    print(f"{rel}: {len(ex) - len(miss)} of {len(ex)} executable lines executed" + (f"; never: {miss}" if miss else ""))
print(f"package: {hit} of {tot} = {100.0 * hit / tot:.1f} %")
