#!/usr/bin/env python3
"""Development: which kernels of libalproj_hip.so did a profiled run launch?  Compares the kernel symbols of the library's gfx950
code object with the kernel names in a rocprofv3 --kernel-trace database.
   rocprofv3 --kernel-trace -d gpurun_out/cov -o p -- python3 -m pytest tests -m gpu -q
   python3 tools/kernel_coverage.py gpurun_out/cov/p_results.db"""
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "alproj_amd", "libalproj_hip.so")
db = sqlite3.connect(sys.argv[1])
ran = {r[0] for r in db.execute("select distinct name from kernels")}
ran_base = {re.sub(r"\(.*", "", n).replace("void ", "").strip() for n in ran}
# the kernels of the last build: the kernel-resource-usage remarks hipcc left in build/*.o.log (alproj_amd/_build.py)
sys.path.insert(0, ROOT)
from alproj_amd import _build                                   # noqa: E402
mangled = sorted(_build.resource_usage())
dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.splitlines()
names = {re.sub(r"\(.*", "", d).replace("void ", "").strip() for d in dem}
ours = {n for n in names if n.startswith("alp::")}
missed = sorted(n for n in ours if n not in ran_base)
print(f"{len(ours)} alp:: kernels in the library, {len(ours) - len(missed)} launched in this run, {len(missed)} not:")
for n in missed:
    print("   ", n)
