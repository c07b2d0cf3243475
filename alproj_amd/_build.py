"""Build recipe for libalproj_hip.so (hipcc, gfx950 only, in-tree).

Used by ``__graft_entry__.build()`` and ``python -m alproj_amd._build``.  The shared
library is written next to this file so that it travels with the source tree; it is
git-ignored (history stays source-only).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
BUILD = os.path.join(ROOT, "build")
LIB = os.path.join(HERE, "libalproj_hip.so")
SOURCES = ["alp_core.hip", "alp_points.hip", "alp_raster.hip", "alp_mesh.hip", "alp_rasterize.hip", "alp_sampler.hip",
           "host/alp_host.cpp"]
# host/: the HIP-free part of the library (plain C++; build_host() compiles the same files with g++ under the sanitizers)
HOST_DIR = os.path.join(CSRC, "host")
HOST_SAN = os.path.join(BUILD, "host_san")
HOST_KINDS = {"plain": [], "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], "tsan": ["-fsanitize=thread"]}
# the raster's coverage/visibility arithmetic is specified operation by operation (DESIGN.md
# section 5): no implicit fused multiply-adds there
EXTRA_FLAGS = {"alp_raster.hip": ["-ffp-contract=off"], "alp_mesh.hip": ["-ffp-contract=off"]}
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libalproj_hip.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libalproj_hip.so."""
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
              [os.path.join(HOST_DIR, f) for f in sorted(os.listdir(HOST_DIR)) if f.endswith(".h")] + \
              [os.path.join(INCLUDE, "alproj_hip.h"), os.path.abspath(__file__)]
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
             f"-I{INCLUDE}", f"-I{CSRC}"]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        host_only = src.endswith(".cpp")
        o = os.path.join(BUILD, os.path.basename(src).replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            if host_only:       # no device code: plain C++ through the same driver (one C++ runtime in the library)
                cmd = [hipcc(), "-x", "c++"] + flags[1:] + ["-c", s, "-o", o]
            else:
                cmd = [hipcc()] + flags + EXTRA_FLAGS.get(src, []) + \
                      ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            with open(o + ".log", "w") as f:
                f.write(r.stderr)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise RuntimeError(f"hipcc failed on {src}")
    if force or _stale(LIB, objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


def host_compiler(name):
    """``clang``: the clang++ hipcc drives (the library's own compiler); ``gcc``: g++"""
    if name == "gcc":
        return shutil.which("g++")
    exe = "/opt/rocm/lib/llvm/bin/clang++"
    return exe if os.path.exists(exe) else shutil.which("amdclang++")


def build_host(kind="plain", compiler="clang", force=False):
    """The HIP-free part of the library (csrc/host/) + its self-checking driver, compiled without HIP: ``plain``,
    ``asan`` (-fsanitize=address,undefined, every report fatal) or ``tsan`` (-fsanitize=thread), by the library's own
    clang++ or by g++.  Returns the driver executable build/host_san/alp_host_<kind>_<compiler>;
    tests/test_host_sanitized.py runs it."""
    cxx = host_compiler(compiler)
    if not cxx:
        raise RuntimeError(f"no {compiler} C++ compiler found")
    os.makedirs(HOST_SAN, exist_ok=True)
    srcs = [os.path.join(HOST_DIR, f) for f in ("alp_host.cpp", "alp_host_selfcheck.cpp")]
    deps = srcs + [os.path.join(HOST_DIR, "alp_host.h"), os.path.join(INCLUDE, "alproj_hip.h"), os.path.abspath(__file__)]
    exe = os.path.join(HOST_SAN, f"alp_host_{kind}_{compiler}")
    if force or _stale(exe, deps):
        cmd = [cxx, "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-Wall", "-Wextra", f"-I{INCLUDE}", f"-I{CSRC}"] + \
              HOST_KINDS[kind] + srcs + ["-o", exe, "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError(f"{cxx} failed on the host-only build ({kind})")
    return exe


def resource_usage():
    """Parse the kernel-resource-usage remarks of the last build: {kernel: {field: value}}."""
    out = {}
    for src in SOURCES:
        log = os.path.join(BUILD, os.path.basename(src).replace(".hip", ".o.log").replace(".cpp", ".o.log"))
        if not os.path.exists(log):
            continue
        cur = None
        for line in open(log):
            if "remark:" not in line or "[-Rpass-analysis" not in line:
                continue
            body = line.split("remark:", 1)[1].split("[-Rpass")[0].strip()
            if body.startswith("Function Name:"):
                cur = body.split(":", 1)[1].strip()
                out[cur] = {}
            elif cur and ":" in body:
                k, v = body.split(":", 1)
                out[cur][k.strip()] = v.strip()
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    for k, v in resource_usage().items():
        print(k, {a: v[a] for a in v if a in ("VGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]",
                                               "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")})
