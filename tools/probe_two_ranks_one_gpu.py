"""Development: can two ranks of an RCCL communicator share the one GPU of a test box?  (If RCCL allows it, the
all-reduce / broadcast paths get a real nranks = 2 execution; it usually refuses with 'duplicate GPU'.)
   python3 tools/probe_two_ranks_one_gpu.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    d = tempfile.mkdtemp()
    ps = [subprocess.Popen([sys.executable, __file__, str(r), os.path.join(d, "uid")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in ps:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate()
            out += b"\n[timeout]"
        print(out.decode()[-1500:])
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
from alproj_amd import _lib as L, dist as adist
rank, path = int(sys.argv[1]), sys.argv[2]
try:
    adist.init_from_file(path, rank, 2, device=0, timeout_s=60)
    print(rank, "comm", L.comm_info(), flush=True)
    x = np.array([float(rank + 1)] * 4)
    L.comm_bcast(x, root=0)
    print(rank, "bcast ->", x, flush=True)
    L.comm_destroy()
except Exception as e:
    print(rank, "FAILED:", e, flush=True)
