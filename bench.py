#!/usr/bin/env python3
"""bench.py -- headline benchmark of the alproj camera-projection hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            # N = 1, 2, 4, 8: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          # ... or takes the ranks a launcher started

With --gpus N > 1 and no WORLD_SIZE in the environment the process is the LAUNCHER (alproj_amd/launch.py): it
never touches the GPU, starts N fresh children of this script (RANK / LOCAL_RANK / WORLD_SIZE set, no exec),
serves their control plane (barrier, max, the 128-byte RCCL id) on a localhost socket, ends the others when one
rank fails, and exits with the job's code.  No torch anywhere: the one collective of the data path is the RCCL
all-reduce inside libalproj_hip.so.  `--launch-selftest` runs the same launch with a stub worker (no GPU).

Workload (BASELINE.json metric: "Gpoints/s projected + CMA-ES iters/s, 100M-vertex DSM,
1/2/4/8 MI355X"): the 100 M-vertex synthetic DSM of SURVEY.md 8(d), resident in HBM as
float32 planes, row-sharded over the ranks (strong scaling: the total is fixed).

  leg 1 (the timed "steps" -> `value`): one step = one single-pose forward projection of the
        whole DSM (every rank projects its shard, no collective).  Gpoints/s = vertices / t.
  `cma`: CMA-ES generations = ask -> population evaluation on the GPU(s) with ONE RCCL all-reduce of
        the P+1 partial sums -> tell; pop = 2048, D = 21 (BASELINE config 5); kernel / all-reduce
        split from HIP events inside the library.
  `f64` (1 GPU): the same projection and population evaluation in the float64 parity mode.
  `c2_c3_10m` (1 GPU): BASELINE configs 2 and 3: 10 M vertices, single-pose projection and CMA-ES
        pop 256 / D 9 (Huber f = 10 and mean distance).
  `raster` (1 GPU only -- the render does not shard: "replicas only"): depth-buffered render of the
        100 M-vertex DSM as a triangle mesh onto the 5616x3744 frame (BASELINE config 4): implicit
        grid, int32 index array, distorted pose, and upload-inclusive.
  `cpu_baseline` (rank 0, N=1 only): the numpy float64 restatement of the reference
        (oracle/ref_numpy.py) timed on this host on a bounded sample.

  `roofline.traffic` (N=1): measured IN the run -- after the legs a child process runs the headline kernel on a DSM of the same
        size under `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes; `--no-live-traffic` or a failure:
        the committed summary's figure, and the line says which).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
VALU_PEAK = 157.3e12       # flop/s fp32 vector (MI355X_MICROARCH.md)
VALU_PEAK_F64 = 78.6e12    # flop/s fp64 vector (MI355X_MICROARCH.md)
BYTES_PER_VERTEX = {"f32": 20, "f64": 40}     # 3 coordinates in + 2 pixel coordinates out
EVAL_FLOPS = 76            # flop per point-candidate evaluation (Huber): 30 fma + 12 + 4 transcendental (+ 1/6 multiply), DESIGN.md section 4 (K2)
# ... and of the lens-free variant (every candidate has k = p = s = 0: the reference's first phase, BASELINE config 3): 9 fma (rows)
# + reciprocal + 2 fma (residuals) + mul + fma (squared distance) + sqrt + min + 2 fma (Huber) = 32; the mean distance needs an add
# instead of min + 2 fma: 28.  The general arithmetic on the same population: 76 / 72.
EVAL_FLOPS_BY_VARIANT = {("lens_free", "huber"): 32, ("lens_free", "mean"): 28, ("general", "huber"): 76, ("general", "mean"): 72,
                         ("shared_pose", "huber"): 76, ("shared_pose", "mean"): 72}
PROFILES = os.path.join(ROOT, "profiles")
ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")      # committed counter summaries: the newest round wins

# The driver's record keeps the first 24 keys of `roofline`: exactly these, in this order, are emitted there -- the five BASELINE
# configs (c2-sized headline, c5-shaped CMA-ES, c3, c4) and the float64 mode; everything else goes to `roofline_detail`
# (tests/test_bench_record.py holds the cap and the names).
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                 "scaled_1e-5_pass", "strict_1e-5_relative_pass",
                 "cma_iters_per_s", "cma_kernel_ms", "cma_valu_frac", "cma_all_reduce_ms",
                 "f64_gpoints_per_s", "f64_hbm_frac", "f64_strict_1e-5_pass",
                 "c3_iters_per_s", "c3_kernel_ms", "c3_valu_frac",
                 "raster_ms_per_frame", "raster_hbm_frac", "raster_binding_roof_frac", "raster_int32_indices_ms_per_frame")
ROOFLINE_CAP = 24


def driver_roofline(flat):
    """Split what the legs measured into (`roofline`: exactly ROOFLINE_KEYS in that order, None for a leg that did not run;
    `roofline_detail`: every other key)."""
    assert len(ROOFLINE_KEYS) <= ROOFLINE_CAP
    roof = {k: flat.get(k) for k in ROOFLINE_KEYS}
    return roof, {k: v for k, v in flat.items() if k not in roof}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--vertices", type=int, default=100_000_000)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--pop", type=int, default=2048)
    ap.add_argument("--dims", type=int, default=21, choices=[9, 21])
    ap.add_argument("--cma-steps", type=int, default=50, help="generations timed at pop 2048 / D 21 (SURVEY 8(d) c5: 50, whatever --steps is)")
    ap.add_argument("--c3-steps", type=int, default=100, help="generations timed at 10 M x pop 256 (SURVEY 8(d) c3: 100)")
    ap.add_argument("--no-cma", action="store_true")
    ap.add_argument("--no-raster", action="store_true")
    ap.add_argument("--no-raster-explicit", action="store_true", help="skip the int32 index-array mesh (2.4 GB of indices)")
    ap.add_argument("--no-f64", action="store_true")
    ap.add_argument("--no-10m", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the reference-typed sim_image + reverse_proj call pair (9.6 GB of host arrays)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the SURVEY 8(f) rows f1-f4 and the full-size pipeline")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes of a child process, ~40 s): take the committed summary's")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="wall-clock limit of a self-launched multi-rank job (s)")
    ap.add_argument("--debug-share-device", nargs="?", const="nocomm", default=None, choices=["nocomm", "rccl"],
                    help="DEVELOPMENT: all ranks use device 0.  'nocomm' (default): no RCCL communicator is made (RCCL refuses two ranks "
                         "on one GPU): walks the multi-rank control flow of this script on a 1-GPU box; the line it prints is marked and "
                         "means nothing.  'rccl': ncclCommInitRank IS called and fails -- shows what a failed RCCL set-up prints and "
                         "that every rank ends")
    ap.add_argument("--debug-comm-world1", action="store_true",
                    help="DEVELOPMENT (N = 1): make a REAL RCCL communicator of one rank, so that every population evaluation goes "
                         "through ncclAllReduce on the library stream -- all a 1-GPU box can show of the collective's cost inside the loop")
    ap.add_argument("--debug-fail-comm-init-rank", type=int, default=-1,
                    help="DEVELOPMENT: this rank gives up right before alp_comm_init (its peers are then inside ncclCommInitRank "
                         "waiting for it): the launcher must end them")
    ap.add_argument("--launch-selftest", action="store_true", help="run the launch / control plane with a stub worker (no GPU, no library)")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="selftest: this rank exits with code 7 after the first barrier")
    ap.add_argument("--selftest-hang-rank", type=int, default=-1, help="selftest: this rank never reaches the second barrier")
    return ap.parse_args()


def timed(ctl, L, fn, steps, warmup):
    """warmup, then EXACTLY `steps` calls bracketed by barrier + device sync on both sides.
    Returns (wall seconds, max over ranks; device ms between HIP events on the library stream)."""
    for _ in range(warmup):
        fn()
    L.synchronize()
    ctl.barrier()
    t0 = time.perf_counter()
    L.event_record(0)
    for _ in range(steps):
        fn()
    L.event_record(1)
    L.synchronize()
    t1 = time.perf_counter()           # this rank's K steps are complete on its GPU
    ctl.barrier()
    dev_ms = L.event_elapsed_ms(0, 1)
    return ctl.max(t1 - t0), dev_ms    # MAX over ranks


def launch_times(L, fn, launches=24):
    """device ms of `launches` single calls, one HIP-event pair each on the library stream (slots 2 ... 2 + launches)"""
    launches = min(launches, 50)
    L.synchronize()
    L.event_record(2)
    for i in range(launches):
        fn()
        L.event_record(3 + i)
    L.synchronize()
    return np.array([L.event_elapsed_ms(2 + i, 3 + i) for i in range(launches)])


def selftest(ctl, args, emit=print):
    """The stub worker of --launch-selftest: what a rank does with the control plane around the real benchmark --
    rendezvous of a 128-byte id from rank 0, barriers, a max over ranks, a gather -- without GPU or library."""
    import hashlib
    print(f"selftest: rank {ctl.rank} pid {os.getpid()}", file=sys.stderr, flush=True)
    os.write(1, b"selftest: a library writes to file descriptor 1 (as RCCL does with its banner): this must not reach stdout\n")
    uid = ctl.bcast_bytes(bytes((7 * i + 1) % 256 for i in range(128)) if ctl.rank == 0 else b"\0" * 128)
    ctl.barrier()
    if ctl.rank == args.selftest_fail_rank:
        print(f"selftest: rank {ctl.rank} fails on purpose", file=sys.stderr, flush=True)
        os._exit(7)
    if ctl.rank == args.selftest_hang_rank:
        time.sleep(1e6)
    ctl.barrier()
    worst = ctl.max(10.0 + ctl.rank)
    recs = ctl.gather({"rank": ctl.rank, "local_rank": ctl.local_rank, "world": ctl.world, "pid": os.getpid(), "ppid": os.getppid(),
                       "uid_sha256": hashlib.sha256(uid).hexdigest(), "hub": "parent" if "ALPROJ_HUB" in os.environ else "rank0",
                       "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}})
    ctl.close()
    if ctl.rank == 0:
        emit(json.dumps({"selftest": True, "n_gpus": ctl.world, "max_over_ranks": worst, "ranks": recs}))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def blas_threads():
    """threads of the BLAS behind numpy's np.dot (the only multi-threaded part of the reference path)"""
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        if pools:
            return {"library": pools[0].get("internal_api"), "threads": pools[0].get("num_threads")}
    except Exception:
        pass
    return {"library": None, "threads": os.environ.get("OPENBLAS_NUM_THREADS") or os.environ.get("OMP_NUM_THREADS")}


def cpu_baseline(orc, truth, base, targets, bounds_fn, xyz_l, obs, n_total, pop):
    """the numpy float64 port of the reference path (oracle = checker only) on the host cores, warm
    best of 3 at N = 1e5, 1e6, 1e7 (strided samples of the same DSM); a population of P = 4 candidates through the
    reference's per-candidate loop (optimize.py:347-356: denormalise, project, huber) extrapolated to P."""
    sizes = {}
    n_local = len(xyz_l)
    for ns in (100_000, 1_000_000, 10_000_000):
        ns = min(ns, n_local)
        st = max(1, n_local // ns)
        xyz = xyz_l[0:ns * st:st].astype(np.float64)
        o = obs[0:ns * st:st].astype(np.float64) if obs is not None else np.zeros((ns, 2))
        tp, tl = [], []
        for i in range(4):                           # one warm-up, then best of 3
            t = time.perf_counter()
            uv = orc.project_points(xyz, truth)
            t1 = time.perf_counter()
            orc.huber(o, uv, 10.0)
            t2 = time.perf_counter()
            if i:
                tp.append(t1 - t)
                tl.append(t2 - t1)
        sizes[ns] = {"project_s": min(tp), "huber_s": min(tl), "gpoints_per_s": ns / min(tp) / 1e9}
        if ns == 1_000_000 or ns == n_local:
            bounds = bounds_fn(base, targets)
            X = np.random.default_rng(4).uniform(0.45, 0.55, (4, len(targets)))
            tpop = []
            for i in range(3):
                t = time.perf_counter()
                with np.errstate(all="ignore"):
                    orc.population_losses(xyz, o, base, targets, bounds, X, 10.0)
                tpop.append(time.perf_counter() - t)
            pop4 = {"points": ns, "candidates": 4, "seconds": min(tpop[1:])}
    big = max(sizes)
    per_eval = pop4["seconds"] / 4 / pop4["points"]             # seconds per point-candidate
    return {
        "value": sizes[big]["gpoints_per_s"], "unit": "Gpoints/s", "cores": 1, "kind": "port",
        "sample": f"numpy float64 restatement of the reference (oracle/ref_numpy.project_points) on strided samples of the same DSM, "
                  f"N = {sorted(sizes)}, warm best of 3; `value` is the N = {big} figure; elementwise numpy is single-threaded, "
                  f"only the two np.dot use BLAS threads",
        "by_n": {str(k): v for k, v in sizes.items()},
        "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "blas": blas_threads(),
        "population_sample": pop4,
        "cma_iters_per_s_extrapolated": 1.0 / (per_eval * n_total * pop),
        "cma_extrapolation": f"P = 4 candidates x {pop4['points']} points through the per-candidate loop, scaled to {pop} x {n_total}",
    }


TOLERANCES = {
    # north_star: "outputs match the reference ... to 1e-5 relative".  The two readings the record reports:
    "strict": "|got - ref| <= 1e-5 * |ref|  (north_star as worded)",
    "scaled": "|got - ref| <= 1e-5 * max(|ref|, image width)  (relative to the size of the image)",
}


def parity_report(got, ref, w):
    """A projection against the float64 oracle under BOTH readings of north_star's "1e-5 relative" (TOLERANCES).
    float64 mode meets the strict one on every value.  float32 meets the scaled one on every value and the strict one on
    all but the small |ref|: a float32 COORDINATE at distance D is uncertain by D * 2^-24, which moves a pixel by up to
    fx * 2^-24 ~ 2e-4 px whatever the arithmetic -- more than 1e-5 |ref| below |ref| ~ 20 px."""
    d = np.abs(got - ref)
    fin = np.isfinite(ref) & np.isfinite(got)
    strict = float((d[fin] <= 1e-5 * np.abs(ref[fin])).mean())
    scaled = float((d[fin] <= 1e-5 * np.maximum(np.abs(ref[fin]), w)).mean())
    return {"checked_values": int(fin.sum()),
            "definitions": TOLERANCES,
            "strict_1e-5_relative_pass_fraction": strict,
            "scaled_1e-5_pass_fraction": scaled,
            "meets": "strict" if strict == 1.0 else ("scaled" if scaled == 1.0 else "neither"),
            "max_err_rel_to_max(|ref|,w)_vs_f64_oracle": float((d[fin] / np.maximum(np.abs(ref[fin]), w)).max()),
            "tolerance": 1e-5,
            "max_abs_err_px": float(d[fin].max())}


def committed_summary(name):
    """(json, path) of the newest committed profiles/rNN_<name>.json"""
    for rnd in ROUNDS:
        f = os.path.join(PROFILES, f"{rnd}_{name}.json")
        if os.path.exists(f):
            try:
                return json.load(open(f)), os.path.relpath(f, ROOT)
            except (OSError, ValueError):      # an unreadable summary must not stop the measurement
                continue
    return None, None


def expectation_from_one_gpu(n_gpus, measured_all_reduce_ms, vertices=None, population=None):
    """For N > 1: what the committed N = 1 run predicts for this N -- kernel time / N (rows are sharded evenly, strong scaling)
    plus the all-reduce THIS run measured, the host share of a generation unchanged -- so that the first scaling curve can be
    read at a glance.  None without a committed N = 1 line."""
    one, src = committed_summary("bench_default_run")
    if not one or one.get("n_gpus") != 1:
        return None
    if vertices is not None and one.get("config", {}).get("vertices") != vertices:
        return None                      # another workload than the committed line's: nothing to compare with
    if population is not None and (one.get("cma") or {}).get("population") not in (None, population):
        return None
    r = dict(one.get("roofline_detail", {}), **one.get("roofline", {}))
    exp = {"source": src, "n_gpus": n_gpus, "rule": "kernel_ms(N=1) / N + all-reduce measured here; host share of a generation as at N = 1"}
    if r.get("kernel_ms"):
        exp["projection_kernel_ms"] = r["kernel_ms"] / n_gpus
        exp["projection_gpoints_per_s"] = one["config"]["vertices"] / (r["kernel_ms"] / n_gpus / 1e3) / 1e9
    cma = one.get("cma") or {}
    if r.get("cma_kernel_ms") and cma.get("ms_per_iter"):
        host_ms = max(0.0, cma["ms_per_iter"] - r["cma_kernel_ms"])
        exp["cma_kernel_ms"] = r["cma_kernel_ms"] / n_gpus
        exp["cma_host_ms_per_iter"] = host_ms
        exp["cma_ms_per_iter"] = r["cma_kernel_ms"] / n_gpus + (measured_all_reduce_ms or 0.0) + host_ms
        exp["cma_iters_per_s"] = 1e3 / exp["cma_ms_per_iter"]
    return exp


def pmc_traffic(name):
    """HBM bytes per launch measured with rocprofv3 --pmc in separate passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, as MI355X_MICROARCH.md prescribes); the newest committed round wins."""
    for rnd in ROUNDS:
        f = os.path.join(PROFILES, f"{rnd}_{name}_pmc_traffic.json")
        if os.path.exists(f):
            try:
                return json.load(open(f)), os.path.relpath(f, ROOT)
            except (OSError, ValueError):      # an unreadable summary must not stop the measurement
                continue
    return None, None


PROFILER_ENV_MARKS = ("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTRACER_")


def under_a_profiler(env=None):
    """is this process itself running under rocprofv3 / rocprof?  (its tool library is preloaded and its ROCP_* variables are
    inherited by every child: a second profiler nested inside would count into the first one's output, or fail)"""
    env = os.environ if env is None else env
    if any(k.startswith(PROFILER_ENV_MARKS) for k in env):
        return True
    return any(w in env.get("LD_PRELOAD", "") for w in ("rocprofiler", "roctracer", "rocprof"))


def live_traffic(vertices, timeout_s=170.0):
    """roofline.traffic MEASURED in this run: HBM bytes per launch of the headline kernel from the PMC counters, collected as
    MI355X_MICROARCH.md prescribes -- two separate `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, doubled on gfx950;
    WRITE_SIZE) -- over a CHILD process (tools/probe_project.py: the same kernel on a DSM of the same size; this process
    cannot put itself under the profiler).  Returns (bytes per launch, source) or (None, why not): the caller then falls back
    on the committed summary and the line names the reason.  The child is ended by PID if it outlives the limit.

    Nothing is written inside the checkout (it may be read-only): the profiler's output and the summary live in ONE
    tempfile.mkdtemp() directory outside the tree, removed on every way out -- success, timeout, a failing pass."""
    import shutil
    import subprocess
    import tempfile
    if under_a_profiler():
        return None, "this run is itself under a profiler (ROCP_* / rocprofiler preload in the environment): no nested PMC pass"
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    tool, probe = os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(ROOT, "tools", "probe_project.py")
    if not (os.path.exists(tool) and os.path.exists(probe)):
        return None, "tools/pmc_traffic.py or tools/probe_project.py missing"
    try:
        work = tempfile.mkdtemp(prefix="alproj_pmc_")            # $TMPDIR or /tmp: never the checkout
    except OSError as e:
        return None, f"no writable temporary directory ({e})"
    try:
        out_json = os.path.join(work, "traffic.json")
        cmd = [sys.executable, tool, out_json, "3", "project_kernel", "--", sys.executable, probe, str(vertices), "3", "f32"]
        env = dict(os.environ, TMPDIR=work, PMC_TRAFFIC_DIR=work, PYTHONDONTWRITEBYTECODE="1")
        p = subprocess.Popen(cmd, cwd=work, env=env, stdin=subprocess.DEVNULL,
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, 15)                 # exactly the process group this call started
            except OSError:
                pass
            p.wait(timeout=20)
            return None, f"the PMC passes did not finish within {timeout_s:.0f} s"
        if rc != 0:
            return None, f"tools/pmc_traffic.py exited with {rc}"
        doc = json.load(open(out_json))
        k = [v for name, v in doc["kernels"].items() if "project_kernel<float>" in name]
        if not k or not k[0].get("hbm_bytes_per_frame"):
            return None, "no counters for project_kernel<float> in the passes"
        return float(k[0]["hbm_bytes_per_frame"]), ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE (x 2 on gfx950) and WRITE_SIZE, separate "
                                                    f"passes over tools/probe_project.py {vertices} 3 f32 (a child process, the same kernel and size)")
    except Exception as e:                       # the measurement must never take the line down with it
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cma_loop(L, CMA, pts, base, targets, bounds_fn, pop, loss_kind, f_scale, seed=1234):
    """One CMA-ES optimiser over the resident point set: returns (generation closure, state dict).  A generation is what
    CMAOptimizer.optimize does for every generation but its last: ask, evaluate (losses; the argmin with its float64
    confirmation of near ties is only needed -- and only asked for -- after the LAST generation, optimize.py:427), tell."""
    bounds = bounds_fn(base, targets)
    lower, upper = bounds[:, 0], bounds[:, 1]
    cols = [L.PARAM_KEYS.index(t) for t in targets]
    basev = L.params_vector(base)
    opt = CMA(mean=np.full(len(targets), 0.5), sigma=1.0,
              bounds=np.column_stack([np.zeros(len(targets)), np.ones(len(targets))]),
              population_size=pop, n_max_resampling=100, seed=seed, sampler=L.cma_sample)      # as CMAOptimizer.optimize does
    state = {"t_ask": 0.0, "t_eval": 0.0, "t_tell": 0.0}

    def generation():
        t0 = time.perf_counter()
        X = opt.ask_population()
        cand = np.tile(basev, (pop, 1))
        cand[:, cols] = X * (upper - lower) + lower
        t1 = time.perf_counter()
        losses, amin = pts.eval_population(cand, loss_kind, f_scale, want_argmin=False)
        t2 = time.perf_counter()
        opt.tell_population(X, losses)
        t3 = time.perf_counter()
        state["best"] = float(losses[amin])
        state["t_ask"] += t1 - t0
        state["t_eval"] += t2 - t1
        state["t_tell"] += t3 - t2
        state["n"] = state.get("n", 0) + 1

    return generation, state


def best_of(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t)
    return best


def dropin_leg(L, syn, surf, n_side, cam):
    """The reference's own call pair with the reference's own array types (example.py:28,31; project.py:213-215):
    sim_image(vert f64, col f64, ind int64, params, offsets), then reverse_proj(sim, vert, ind, params, offsets) --
    as the reference runs it (the mesh uploaded on every call: the default), and with the opt-in mesh cache: writeable
    arrays (every byte digested before the resident mesh is trusted) and read-only arrays (taken by identity)."""
    from alproj_amd import project as aproj
    n_total = n_side * n_side
    vert64 = surf["vert"].astype(np.float64)             # get_colored_surface returns float64 (surface.py:189-193)
    col64 = np.random.default_rng(syn.SEED + 1).random((n_total, 3))
    ind64 = syn.grid_indices(n_side, np.int64)           # 4.8 GB (docs/usage.md:96)
    nbytes = vert64.nbytes + col64.nbytes + ind64.nbytes
    aproj.set_timing(True)
    rec = {"call": "sim_image(vert f64, col f64, ind int64, params, offsets); reverse_proj(sim, vert, ind, params, offsets)",
           "vertices": n_total, "host_bytes_of_the_three_arrays": nbytes}
    pcie = 56e9                                          # B/s, measured pageable H2D rate of this box class (tools/h2d_rate.hip)

    def pair(tag):
        t = time.perf_counter()
        sim = aproj.sim_image(vert64, col64, ind64, cam, surf["offsets"])
        t_sim = time.perf_counter() - t
        r = {"sim_image": dict(aproj.LAST_TIMING, total_s=t_sim)}
        t = time.perf_counter()
        df = aproj.reverse_proj(sim, vert64, ind64, cam, surf["offsets"])
        t_rev = time.perf_counter() - t
        r["reverse_proj"] = dict(aproj.LAST_TIMING, total_s=t_rev, rows=len(df))
        r["first_call_ms"], r["second_call_ms"] = t_sim * 1e3, t_rev * 1e3
        r["second_call_device_ms"] = r["reverse_proj"].get("device_ms")
        r["second_call_resolve_only"] = bool(r["reverse_proj"].get("resolve_only"))
        rec[tag] = r
        return sim, df

    # (a) as the reference does it: every call uploads
    sim, df = pair("default_upload_every_call")
    rec["first_call_ms"] = rec["default_upload_every_call"]["first_call_ms"]
    rec["first_call_pcie_floor_ms"] = nbytes / pcie * 1e3
    rec["first_call_over_pcie_floor"] = rec["first_call_ms"] / rec["first_call_pcie_floor_ms"]
    # (b) opt-in cache, writeable arrays: a hit costs a digest of every byte (alp_host_hash64, all host cores)
    t = time.perf_counter()
    L.host_hash64(vert64), L.host_hash64(col64), L.host_hash64(ind64)
    t_dig = time.perf_counter() - t
    rec["digest_of_the_three_arrays_ms"] = t_dig * 1e3
    rec["digest_gb_per_s"] = nbytes / t_dig / 1e9
    aproj.set_mesh_cache(True)
    sim_b, df_b = pair("verify")
    assert np.array_equal(sim, sim_b) and df_b.equals(df)
    del df_b, sim_b
    # (c) opt-in cache, read-only arrays: identity is enough
    aproj.clear_mesh_cache()
    for a in (vert64, col64, ind64):
        a.setflags(write=False)
    sim_c, df_c = pair("read_only_arrays")
    assert np.array_equal(sim, sim_c) and df_c.equals(df)
    t = time.perf_counter()
    sim2 = aproj.sim_image(vert64, col64, ind64, cam, surf["offsets"])
    rec["read_only_arrays"]["sim_image_again"] = dict(aproj.LAST_TIMING, total_s=time.perf_counter() - t)
    assert np.array_equal(sim, sim2)
    del df_c, sim_c, sim2
    rec["second_call_device_ms"] = rec["read_only_arrays"]["second_call_device_ms"]
    rec["second_call_resolve_only"] = rec["read_only_arrays"]["second_call_resolve_only"]
    # what the reference does on the host before its own upload (project.py:213-215), same box, one core (numpy)
    t = time.perf_counter()
    a, b, c = vert64.astype("f4"), col64.astype("f4"), ind64.astype("i4")
    rec["reference_host_casts_s"] = time.perf_counter() - t
    del a, b, c
    aproj.set_mesh_cache(False)
    aproj.set_timing(False)
    return rec, df, sim


def next_rows_leg(L, syn, orc, df, ras_dev=None):
    """kernel time from HIP events inside the library (alp_kernel_timing), algorithmic bytes
    over it against the HBM peak, the numpy port of the reference timed beside it on a bounded sample."""
    from alproj_amd import project as aproj
    out = {}
    L.kernel_timing(True)
    # ---- f1: residuals of D + 1 = 22 poses over 10 M points, one launch per chunk (optimize.py:215-237, 442-539)
    n10 = syn.grid_side(10_000_000)
    s10 = syn.surface(n10)
    x10 = syn.vert_to_xyz_local(s10["vert"])
    b10 = syn.local_params(syn.standoff_params(n10), s10["offsets"])
    t10 = syn.local_params(syn.perturbed(syn.standoff_params(n10)), s10["offsets"])
    n = len(x10)
    B = 22
    with L.Points(x10, [b10["x"], b10["y"], b10["z"]], "f64") as p:
        p.project(L.params_vector(t10))
        u, v = p.fetch()
        p.set_observed(np.stack([u, v], 1) + np.random.default_rng(2).normal(0, 1.0, (n, 2)))
        cand = np.tile(L.params_vector(b10), (B, 1))
        cand[:, 4] += np.linspace(-0.5, 0.5, B)            # pan: 22 distinct poses
        p.residuals_batch(cand)                              # warm-up (scratch, first touch of the output pages)
        L.kernel_time_ms()
        t = time.perf_counter()
        res = p.residuals_batch(cand)
        wall = time.perf_counter() - t
        k_ms, sections = L.kernel_time_ms()
    alg = n * (3 * 8 + 2 * 8) + B * n * 16
    ns = 1_000_000
    t_cpu = best_of(lambda: orc.residual_vector(x10[:ns].astype(np.float64), np.zeros((ns, 2)), t10), 2)
    out["f1_residuals_batch"] = {
        "points": n, "poses": B, "call_ms_incl_3.5GB_fetch": wall * 1e3, "kernel_ms": k_ms, "kernel_launches": sections,
        "pose_point_residuals_per_s_kernel": n * B / (k_ms / 1e3),
        "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (k_ms / 1e3) / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": alg / (k_ms / 1e3) / HBM_PEAK, "kernel": "residual_batch_kernel<double>",
                     # SURVEY prices the row in bytes; what binds it is the float64 arithmetic of B projections per point:
                     "binding_roof": "valu_fp64", "pose_point_evals_per_s": n * B / (k_ms / 1e3),
                     "note": "the float64 population kernel, same stages (norm_coords, distort_group) without the 16 B of residuals written per evaluation, runs at 3.7-3.9e11 evaluations/s"},
        "cpu_port": {"pose_point_residuals_per_s": ns / t_cpu, "sample": f"oracle.residual_vector, {ns} points, 1 pose, 1 core"}}
    del res, x10, s10
    # ---- f3: mesh construction from rasters at 10 000 x 10 000 (surface.py:173-212)
    n3 = 10_000
    rng = np.random.default_rng(3)
    dsm = (1500 + 200 * np.sin(np.arange(n3, dtype=np.float32) / 300.0)[None, :] * np.cos(np.arange(n3, dtype=np.float32) / 200.0)[:, None]).astype(np.float32)
    aerial = rng.integers(0, 256, (3, n3, n3), dtype=np.uint8)
    nodata = np.zeros((n3, n3), dtype=np.uint8)
    nodata[4000:4040, 3000:3600] = 1
    tr = (1.0, 0.0, 732000.0, 0.0, -1.0, 4048000.0 + n3)
    # warm-up on a corner: the first launch of a kernel loads its code object (0.5 ms each), which would sit between
    # the events that bracket the kernel sections
    L.Mesh.from_rasters(dsm[:64, :64], (1.0, 0.0, 0.0, 0.0, -1.0, 64.0), 3000.0, aerial[:, :64, :64], 255.0, nodata[:64, :64])[0].close()
    L.kernel_time_ms()
    t = time.perf_counter()
    mesh3, off3 = L.Mesh.from_rasters(dsm, tr, 3000.0, aerial, 255.0, nodata)
    wall = time.perf_counter() - t
    k_ms, sections = L.kernel_time_ms()
    mesh3.close()
    nv = n3 * n3
    alg = nv * (4 + 3 + 1) + nv * 4 + nv * (12 + 12 + 1)        # build kernel in + z-min pass + vert / value / valid out
    m = 2000
    t_cpu = best_of(lambda: orc.colored_surface(aerial[:, :m, :m], dsm[:m, :m].astype(np.float64), tr, nodata[:m, :m].astype(bool), np.uint8), 2)
    out["f3_mesh_from_rasters"] = {
        "vertices": nv, "call_ms_incl_upload": wall * 1e3, "kernel_ms": k_ms, "kernel_sections": sections,
        "gvertices_per_s_kernel": nv / (k_ms / 1e3) / 1e9,
        "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (k_ms / 1e3) / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": alg / (k_ms / 1e3) / HBM_PEAK, "kernel": "surface_zmin_kernel + surface_build_kernel"},
        "cpu_port": {"gvertices_per_s": m * m / t_cpu / 1e9, "sample": f"oracle.colored_surface on a {m}x{m} corner (incl. the index array the reference builds), 1 core"}}
    del dsm, aerial, nodata
    # ---- f2: to_geotiff's compute on the reverse_proj table of the 100 M-vertex frame (project.py:434-485)
    if df is not None:
        f2 = {"points": len(df), "resolution_m": 1.0, "bands": 3}
        for agg in ("mean", "median"):
            aproj.rasterize(df, 1.0, ["B", "G", "R"], True, 1.0, agg)          # warm-up
            L.kernel_time_ms()
            t = time.perf_counter()
            raster, bounds = aproj.rasterize(df, 1.0, ["B", "G", "R"], True, 1.0, agg)
            wall = time.perf_counter() - t
            k_ms, _ = L.kernel_time_ms()
            total = raster.size
            npts = len(df)
            # round 4 (sort-based, fixed order; byte-valued bands ride the sort as its payload): points in (x, y, 3 values: float64),
            # the values read once more by the byte check and once by the packing, (cell, packed values) written, four 8-bit passes
            # of the pair sort (16 B read + written per pass), the runs read back; per band-cell: one byte filled and (the tiles that
            # hold points: a twelfth of this raster, not counted) a float32 written, read and its byte written again.
            # `frac_at_round3_bytes`: against the 6.4 GB the accumulator design of round 3 was priced at.
            alg = npts * (16 + 8 * 3) + npts * 8 * 3 * 2 + npts * 8 + npts * 16 * 4 + npts * 8 + total if agg == "mean" else None
            f2[agg] = {"call_ms_incl_transfers": wall * 1e3, "kernel_ms": k_ms, "raster": list(raster.shape),
                       "mpoints_per_s_kernel": npts / (k_ms / 1e3) / 1e6}
            if agg == "mean":
                f2[agg]["frac_at_round3_bytes"] = (npts * (16 + 8 * 3) + total * (2 * 12 + 1)) / (k_ms / 1e3) / HBM_PEAK
            if agg == "mean" and ras_dev is not None:
                f2["equals_device_fed_raster"] = bool(np.array_equal(raster, ras_dev))
            if alg:
                f2[agg]["roofline"] = {"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (k_ms / 1e3) / 1e9, "peak": HBM_PEAK / 1e9,
                                       "unit": "GB/s", "frac": alg / (k_ms / 1e3) / HBM_PEAK,
                                       "kernel": "rz_integer_check + rz_pack + rz_cell + rocPRIM radix_sort_pairs + rz_pieces<packed> + rz_join + fill + rz_tail (of the 1.1 ms the sort takes 0.41)"}
        # numpy / pandas port of the reference on a 600 m x 600 m window of the same table (its 3x3 focal pass is a Python lambda per pixel)
        x0, y0 = df["x"].min(), df["y"].median()
        win = df[(df["x"] < x0 + 600) & (np.abs(df["y"] - y0) < 300)]
        if len(win) > 1000:
            t_cpu = best_of(lambda: orc.rasterize_points(win["x"].to_numpy(), win["y"].to_numpy(), win[["B", "G", "R"]].to_numpy(dtype=np.float64),
                                                         1.0, True, 1.0, "mean"), 1)
            f2["cpu_port"] = {"mpoints_per_s": len(win) / t_cpu / 1e6, "sample": f"oracle.rasterize_points (mean) on the {len(win)} rows of a 600 x 600 m window, 1 core"}
        out["f2_rasterize_points"] = f2
    L.kernel_timing(False)
    return out


def main():
    args = parse()
    from alproj_amd import launch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the LAUNCHER: checks that the node has N GPUs (sysfs, no HIP), builds the library once (hipcc only), then starts the
        # ranks; it never initialises the GPU.  The stub worker needs no GPU: its pre-flight runs only against a named tree.
        if not args.debug_share_device and (not args.launch_selftest or os.environ.get(launch.KFD_ENV)):
            if not launch.preflight(args.gpus):
                sys.exit(2)
        if not args.launch_selftest:
            try:
                from alproj_amd import _build
                _build.build()
            except Exception as e:
                print(f"bench.py: build skipped: {e}", file=sys.stderr)
        sys.exit(launch.spawn([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout_s=args.launch_timeout))

    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes its banner and NCCL_DEBUG output to file
    # descriptor 1 (the GPU boxes export NCCL_DEBUG=VERSION), which would land in front of the line as soon as a communicator
    # is made.  From here on descriptor 1 IS stderr; the line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + "\n").encode())

    ctl = launch.Control.from_env()
    if ctl.world != args.gpus:
        if ctl.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ctl.world}", file=sys.stderr)
        sys.exit(2)
    if args.launch_selftest:
        return selftest(ctl, args, emit)

    if ctl.local_rank == 0 and ctl.world == 1:  # harness convenience: (re)build a missing/stale library (the launcher did it for its ranks)
        try:
            from alproj_amd import _build
            _build.build()
        except Exception as e:                 # the product still fails loudly below if it is absent
            print(f"bench.py: build skipped: {e}", file=sys.stderr)
    elif "ALPROJ_HUB" not in os.environ:       # ranks of a foreign launcher: local rank 0 builds, the others wait
        if ctl.local_rank == 0:
            try:
                from alproj_amd import _build
                _build.build()
            except Exception as e:
                print(f"bench.py: build skipped: {e}", file=sys.stderr)
        ctl.barrier()
    from alproj_amd import _lib as L
    from alproj_amd import dist as adist
    from alproj_amd import synthetic as syn
    from alproj_amd.cma import CMA
    from alproj_amd.optimize import bounds_to_array      # the product's own host logic
    from oracle import ref_numpy as orc       # checker (parity spot check) and cpu_baseline only

    if args.debug_share_device == "nocomm":
        L.init(0)
        ctl.bcast_bytes(b"\0" * 128)             # the id's trip is still made
    else:
        def id_trip(b):
            b = ctl.bcast_bytes(b)
            if ctl.rank == args.debug_fail_comm_init_rank:
                print(f"bench.py: rank {ctl.rank} gives up before alp_comm_init (--debug-fail-comm-init-rank)", file=sys.stderr, flush=True)
                os._exit(9)
            return b
        try:
            adist.init_comm(ctl.rank, ctl.world, id_trip, 0 if args.debug_share_device else ctl.local_rank)
        except L.AlprojHipError as e:
            # one block per rank, RCCL's own text included (alp_comm_init); the launcher ends the other ranks
            print(f"bench.py: rank {ctl.rank}/{ctl.world} (LOCAL_RANK {ctl.local_rank}) could not set up: {e}", file=sys.stderr, flush=True)
            sys.exit(5)
    if args.debug_comm_world1 and ctl.world == 1:
        L.comm_init(L.comm_unique_id(), 0, 1)
        print("bench.py: --debug-comm-world1: a one-rank RCCL communicator exists; alp_eval_population all-reduces through it", file=sys.stderr, flush=True)
    info = L.device_info()
    comm_rank, comm_world = L.comm_info()
    if args.debug_share_device == "nocomm":
        comm_rank, comm_world = ctl.rank, ctl.world
    # the communicator the LIBRARY reports must be the job: a rank that fell back to a world of its own would add
    # nothing to the all-reduce and the line would still look plausible
    device = L._device if L._device is not None else (0 if args.debug_share_device else ctl.local_rank)   # where alp_init really went
    print(f"bench.py: rank {ctl.rank}/{ctl.world} on device {device} ({info.get('pci_bus_id')}): rccl rank {comm_rank} of {comm_world}"
          + (" [--debug-share-device: no communicator]" if args.debug_share_device else ""),
          file=sys.stderr, flush=True)
    if comm_world != args.gpus or comm_rank != ctl.rank:
        print(f"bench.py: RCCL communicator has {comm_world} ranks (this one is {comm_rank}) but --gpus is {args.gpus}",
              file=sys.stderr, flush=True)
        sys.exit(3)
    if L.build_flags():
        print(f"bench.py: libalproj_hip.so was built with development switches: {L.build_flags()}", file=sys.stderr)
        sys.exit(4)
    rank_devices = ctl.gather({"rank": ctl.rank, "device": device, "pci_bus_id": info.get("pci_bus_id"),
                               "rccl_rank": comm_rank, "rccl_nranks": comm_world, "pid": os.getpid()})

    # ---------------------------------------------------------------- workload
    n_side = syn.grid_side(args.vertices)
    r0, r1 = adist.shard_rows(n_side, ctl.rank, ctl.world)
    t_gen = time.perf_counter()
    surf = syn.surface(n_side, rows=(r0, r1))       # every rank generates its own rows only
    xyz_l = syn.vert_to_xyz_local(surf["vert"])
    n_local = xyz_l.shape[0]
    n_total = n_side * n_side
    base = syn.local_params(syn.standoff_params(n_side), surf["offsets"])
    truth = syn.local_params(syn.perturbed(syn.standoff_params(n_side)), surf["offsets"])
    origin = [base["x"], base["y"], base["z"]]
    t_up = time.perf_counter()
    pts = L.Points(xyz_l, origin, args.precision)
    t_up = time.perf_counter() - t_up              # host -> device upload of the vertex array (not in `value`)
    t_up2 = time.perf_counter()                    # ... and once more: the first large copy of a process pays the
    L.Points(xyz_l, origin, args.precision).close()   # runtime's one-off set-up of its pageable-memory path
    t_up2 = time.perf_counter() - t_up2
    t_gen = time.perf_counter() - t_gen
    pv_truth = L.params_vector(truth)

    # ---------------------------------------------------------------- leg 1: projection
    wall, dev_ms = timed(ctl, L, lambda: pts.project(pv_truth), args.steps, args.warmup)
    ms_per_step = wall / args.steps * 1e3
    gpts = n_total / (wall / args.steps) / 1e9
    kern_s = dev_ms / args.steps / 1e3
    bpv = BYTES_PER_VERTEX[args.precision]
    achieved = n_local * bpv / kern_s
    per_launch = launch_times(L, lambda: pts.project(pv_truth))      # one event pair per launch: the median next to the mean

    t_fetch = time.perf_counter()
    uu_all, vv_all = pts.fetch(np.float32)          # device -> host of all projected pixels
    t_fetch = time.perf_counter() - t_fetch
    del uu_all, vv_all
    # parity spot check on the bench workload itself (outside the timed region)
    step = max(1, n_local // 40000)
    cnt = (n_local - 1) // step
    u, v = pts.fetch_strided(0, step, cnt)
    sample = xyz_l[0:cnt * step:step].astype(np.float64)
    ref = orc.project_points(sample, truth)
    parity = parity_report(np.stack([u, v], 1), ref, truth["w"])

    traffic, traffic_src = (None, None)
    if args.precision == "f32":
        t_json, traffic_src = pmc_traffic("project")
        if t_json:
            # either the per-vertex summary or tools/pmc_traffic.py's per-kernel table of a 100 M-vertex probe
            per_vertex = t_json.get("hbm_bytes_per_vertex")
            if per_vertex is None and t_json.get("kernels"):
                k = next(iter(t_json["kernels"].values()))
                per_vertex = k["hbm_bytes_per_frame"] / float(t_json.get("vertices_per_launch", 100_000_000))
            traffic = per_vertex * n_local if per_vertex is not None else None

    out = {
        "metric": "Gpoints/s projected (pinhole + Brown-Conrady, single pose) over the 100M-vertex DSM; "
                  "CMA-ES iterations/s in `cma`",
        "value": gpts, "unit": "Gpoints/s", "n_gpus": ctl.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        # (strings of at most 128 characters: the driver's record cuts longer ones)
        "config": {"workload": f"{n_total}-vertex synthetic DSM ({n_side}x{n_side}), single-pose projection, 5616x3744 camera, SoA planes in HBM",
                   "vertices": n_total, "vertices_per_gpu": n_local, "sharding": f"rows/{ctl.world}",
                   "precision": args.precision, "bytes_per_vertex": bpv,
                   # which number meets which reading of north_star's "1e-5 relative" (pass fractions: roofline.*_pass)
                   "tol_strict": ("|d|<=1e-5|ref|: met by the float64 mode (roofline.f64_gpoints_per_s, f64_hbm_frac; 40 B/vertex)" if args.precision == "f32"
                                  else "|d|<=1e-5|ref|: met by `value` (this run: float64, 40 B/vertex): roofline.strict_1e-5_relative_pass"),
                   "tol_scaled": ("|d|<=1e-5 max(|ref|,w): met by `value` (float32, 20 B/vertex); strict: roofline.strict_1e-5_relative_pass" if args.precision == "f32"
                                  else "|d|<=1e-5 max(|ref|,w): met by `value` too (roofline.scaled_1e-5_pass)")},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "traffic_source": f"{traffic_src} (rocprofv3 --pmc, bytes/vertex x vertices per launch)" if traffic else None,
                     "kernel": "project_kernel", "kernel_ms": kern_s * 1e3,
                     "kernel_ms_median_of_single_launches": float(np.median(per_launch)), "single_launches": len(per_launch),
                     "bytes_per_vertex": bpv, "vertices_per_launch": n_local,
                     "scaled_1e-5_pass": parity["scaled_1e-5_pass_fraction"],
                     "strict_1e-5_relative_pass": parity["strict_1e-5_relative_pass_fraction"]},
        "parity": parity,
        "device": info, "setup_s": t_gen,
        # the communicator the library itself reports (ncclCommInitRank succeeded on every rank), and where every rank ran
        "rccl": {"rank": comm_rank, "nranks": comm_world, "ranks": rank_devices} if not args.debug_share_device else
                "NONE: --debug-share-device (all ranks on device 0, no communicator: this line is a control-flow check, not a measurement)",
        # SURVEY 8(d) c2 asks for the end-to-end figure beside the kernel figure; it is never `value`
        "pcie_inclusive": {"upload_s": t_up, "upload_s_second_time": t_up2,
                           "upload_gb_per_s_second_time": xyz_l.nbytes / t_up2 / 1e9, "fetch_uv_s": t_fetch,
                           "gpoints_per_s_one_pass_incl_upload_and_fetch":
                               n_local / (t_up + ms_per_step / 1e3 + t_fetch) / 1e9},
    }

    # ---------------------------------------------------------------- CMA-ES generations (config 5)
    obs = None
    targets = syn.TARGETS_D9 if args.dims == 9 else syn.TARGETS_D21
    if not args.no_cma:
        uu, vv = pts.fetch(np.float32)
        rng = np.random.default_rng(1 + ctl.rank)
        obs = np.stack([uu, vv], 1)
        del uu, vv
        obs += rng.normal(0.0, 1.0, obs.shape).astype(np.float32)
        obs[~np.isfinite(obs)] = 0.0
        pts.set_observed(obs)
        generation, state = cma_loop(L, CMA, pts, base, targets, bounds_to_array, args.pop, L.LOSS_HUBER, 10.0)
        k_cma = args.cma_steps                               # 50 generations, whatever --steps is
        wall_c, _ = timed(ctl, L, generation, k_cma, 2)
        eval_ms, ar_ms = pts.eval_population_timing()        # the last generation's kernels / all-reduce (HIP events)
        eval_ms, ar_ms = ctl.max(eval_ms), ctl.max(ar_ms)    # the slowest rank's (they wait for each other in the all-reduce)
        evals = n_local * args.pop
        n_gen = state["n"]
        out["cma"] = {
            "iters_per_s": k_cma / wall_c, "ms_per_iter": wall_c / k_cma * 1e3, "generations_timed": k_cma,
            "population": args.pop, "dims": len(targets), "loss": "huber f_scale=10",
            "point_candidate_evals_per_s": n_total * args.pop * k_cma / wall_c,
            "best_loss_last_generation": state.get("best"),
            "collective": "ncclAllReduce(sum, f64, P+1) per generation" if ctl.world > 1 else
                          ("ncclAllReduce(sum, f64, P+1) per generation over a ONE-rank communicator (--debug-comm-world1)" if args.debug_comm_world1 else "none (1 GPU)"),
            "all_reduce_ms": ar_ms, "all_reduce_share_of_generation": ar_ms / (wall_c / k_cma * 1e3),
            "host_ms_per_generation": {"ask": state["t_ask"] / n_gen * 1e3, "eval_call": state["t_eval"] / n_gen * 1e3,
                                       "tell": state["t_tell"] / n_gen * 1e3},
            "roofline": {"bound": "valu_fp32", "kernel": "popeval_kernel (+ reduce_partials_kernel)",
                         "kernel_ms": eval_ms,
                         "achieved": evals * EVAL_FLOPS / (eval_ms / 1e3) / 1e12, "peak": VALU_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": evals * EVAL_FLOPS / (eval_ms / 1e3) / VALU_PEAK,
                         "flop_per_eval": EVAL_FLOPS,
                         # SURVEY 8(d) prices the reference's arithmetic at 100 flop per point-candidate
                         # (88 projection + 12 residual / Huber / accumulate); the kernel executes 77
                         "flop_per_eval_survey": 100,
                         "frac_at_survey_flops": evals * 100 / (eval_ms / 1e3) / VALU_PEAK,
                         "hbm_frac": n_local * 20 * ((args.pop + 127) // 128) / (eval_ms / 1e3) / HBM_PEAK},
        }
        # flat copies where the driver keeps them (it stores `roofline` verbatim, other top-level keys by name only)
        out["roofline"].update({"cma_iters_per_s": k_cma / wall_c, "cma_kernel_ms": eval_ms, "cma_generations_timed": k_cma,
                                "cma_population": args.pop, "cma_dims": len(targets),
                                "cma_valu_frac": out["cma"]["roofline"]["frac"], "cma_valu_frac_at_survey_100_flop": out["cma"]["roofline"]["frac_at_survey_flops"],
                                "cma_all_reduce_ms": ar_ms, "cma_all_reduce_share": out["cma"]["all_reduce_share_of_generation"]})

        # the reference's FIRST phase at this scale (example.py:51-54: targets x, y, z, fov, pan, tilt, roll, a1, a2 around a camera
        # without lens coefficients): the lens-free kernel variant, every rank on its rows, the same all-reduce -- ten generations
        if len(targets) == 21 and not any(base.get(k, 0.0) for k in L.DIST_KEYS[2:]):
            gen9, st9 = cma_loop(L, CMA, pts, base, syn.TARGETS_D9, bounds_to_array, args.pop, L.LOSS_HUBER, 10.0)
            wall9, _ = timed(ctl, L, gen9, 10, 1)
            e9, _ = pts.eval_population_timing()
            e9 = ctl.max(e9)
            variant9 = pts.eval_population_info()[0]
            flop9 = EVAL_FLOPS_BY_VARIANT[(variant9, "huber")]
            out["cma"]["d9_first_phase"] = {"iters_per_s": 10 / wall9, "kernel_ms": e9, "generations_timed": 10, "dims": 9, "population": args.pop,
                                            "kernel_variant": variant9, "flop_per_eval": flop9,
                                            "valu_frac": evals * flop9 / (e9 / 1e3) / VALU_PEAK}
            out["roofline"].update({"cma_d9_first_phase_iters_per_s": 10 / wall9, "cma_d9_first_phase_kernel_ms": e9,
                                    "cma_d9_first_phase_kernel_variant": variant9})

    # ---------------------------------------------------------------- float64 parity mode (1 GPU)
    if ctl.world == 1 and not args.no_f64 and args.precision == "f32":
        p64 = L.Points(xyz_l, origin, "f64")
        k64 = max(20, min(args.steps, 40))
        wall64, dev64 = timed(ctl, L, lambda: p64.project(pv_truth), k64, 3)
        u64, v64 = p64.fetch_strided(0, step, cnt)
        par64 = parity_report(np.stack([u64, v64], 1), ref, truth["w"])
        par64["tolerance"] = 1e-9
        k64_s = dev64 / k64 / 1e3
        out["f64"] = {"projection": {"gpoints_per_s": n_total / (wall64 / k64) / 1e9, "ms_per_step": wall64 / k64 * 1e3,
                                     "steps": k64, "parity": par64,
                                     "roofline": {"bound": "hbm", "achieved": n_local * 40 / k64_s / 1e9, "peak": HBM_PEAK / 1e9,
                                                  "unit": "GB/s", "frac": n_local * 40 / k64_s / HBM_PEAK, "kernel_ms": k64_s * 1e3,
                                                  "bytes_per_vertex": 40}}}
        out["roofline"].update({"f64_gpoints_per_s": out["f64"]["projection"]["gpoints_per_s"], "f64_hbm_frac": n_local * 40 / k64_s / HBM_PEAK,
                                "f64_kernel_ms": k64_s * 1e3, "f64_strict_1e-5_pass": par64["strict_1e-5_relative_pass_fraction"],
                                "f64_max_err_rel": par64["max_err_rel_to_max(|ref|,w)_vs_f64_oracle"]})
        if not args.no_cma:
            p64.set_observed(obs)
            gen64, st64 = cma_loop(L, CMA, p64, base, targets, bounds_to_array, args.pop, L.LOSS_HUBER, 10.0)
            k64c = 10                                            # ten generations (two until round 4)
            wall_c64, _ = timed(ctl, L, gen64, k64c, 1)
            e64, _ = p64.eval_population_timing()
            out["f64"]["cma"] = {"iters_per_s": k64c / wall_c64, "ms_per_iter": wall_c64 / k64c * 1e3, "generations_timed": k64c,
                                 "population": args.pop, "dims": len(targets), "kernel_ms": e64,
                                 "point_candidate_evals_per_s": n_total * args.pop * k64c / wall_c64,
                                 "roofline": {"bound": "valu_fp64", "achieved": n_local * args.pop * EVAL_FLOPS / (e64 / 1e3) / 1e12,
                                              "peak": VALU_PEAK_F64 / 1e12, "unit": "TFLOP/s",
                                              "frac": n_local * args.pop * EVAL_FLOPS / (e64 / 1e3) / VALU_PEAK_F64}}
            out["roofline"].update({"f64_cma_iters_per_s": k64c / wall_c64, "f64_cma_kernel_ms": e64,
                                    "f64_cma_valu_fp64_frac": out["f64"]["cma"]["roofline"]["frac"]})
        p64.close()

    # ---------------------------------------------------------------- BASELINE configs 2 and 3: 10 M vertices (1 GPU)
    if ctl.world == 1 and not args.no_10m and n_total > 20_000_000:
        n10 = syn.grid_side(10_000_000)
        s10 = syn.surface(n10)
        x10 = syn.vert_to_xyz_local(s10["vert"])
        b10 = syn.local_params(syn.standoff_params(n10), s10["offsets"])
        t10 = syn.local_params(syn.perturbed(syn.standoff_params(n10)), s10["offsets"])
        with L.Points(x10, [b10["x"], b10["y"], b10["z"]], "f32") as p10:
            pv10 = L.params_vector(t10)
            w10, d10 = timed(ctl, L, lambda: p10.project(pv10), args.steps, args.warmup)
            k10 = d10 / args.steps / 1e3
            pl10 = launch_times(L, lambda: p10.project(pv10))       # kernel only, median of >= 20 launches after warm-up
            med10 = float(np.median(pl10)) / 1e3
            c2 = {"vertices": len(x10), "gpoints_per_s": len(x10) / (w10 / args.steps) / 1e9, "ms_per_step": w10 / args.steps * 1e3,
                  "gpoints_per_s_kernel_median": len(x10) / med10 / 1e9, "kernel_ms_median": med10 * 1e3, "single_launches": len(pl10),
                  "roofline": {"bound": "hbm", "achieved": len(x10) * 20 / k10 / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": len(x10) * 20 / k10 / HBM_PEAK, "kernel_ms": k10 * 1e3, "frac_at_median": len(x10) * 20 / med10 / HBM_PEAK}}
            out["roofline"].update({"c2_gpoints_per_s_kernel_median": c2["gpoints_per_s_kernel_median"], "c2_hbm_frac_at_median": c2["roofline"]["frac_at_median"]})
            # SURVEY 8(d) c2 "end-to-end incl. H2D / D2H": the reference's own call, optimize.project(DataFrame[x, y, z], params) ->
            # DataFrame[u, v] (float64 in, float64 out, the float64 point set that is its default)
            import pandas as pd
            from alproj_amd import optimize as aopt
            df10 = pd.DataFrame(x10.astype(np.float64), columns=["x", "y", "z"])
            aopt.project(df10.iloc[:1000], t10)                        # warm-up of the small things
            t_call = best_of(lambda: aopt.project(df10, t10), 2)
            c2["dropin_project_call"] = {"call": "optimize.project(DataFrame(10 M x 3 float64), params) -> DataFrame[u, v] float64",
                                         "seconds": t_call, "gpoints_per_s": len(df10) / t_call / 1e9,
                                         "pcie_floor_seconds_at_56_GB_per_s": len(df10) * 40 / 56e9,
                                         "reference_seconds_survey_container": 1.98}
            # ... and the reference's stand-alone losses on that result (optimize.py:157-212): two tables in, a float out
            prj10 = aopt.project(df10, t10)
            img10 = pd.DataFrame({"u": prj10["u"].to_numpy() + 0.5, "v": prj10["v"].to_numpy() - 0.25})
            c2["dropin_loss_calls"] = {"rmse_seconds": best_of(lambda: aopt.rmse(img10, prj10), 2),
                                       "huber_loss_seconds": best_of(lambda: aopt.huber_loss(img10, prj10, 10.0), 2),
                                       "pcie_floor_seconds_at_56_GB_per_s": len(df10) * 32 / 56e9,
                                       "reference_seconds_survey_container": {"rmse": 0.159, "huber_loss": 0.262}}
            del df10, prj10, img10
            uu, vv = p10.fetch(np.float32)
            o10 = np.stack([uu, vv], 1) + np.random.default_rng(1).normal(0, 1.0, (len(x10), 2)).astype(np.float32)
            o10[~np.isfinite(o10)] = 0.0
            p10.set_observed(o10)
            c3 = {"vertices": len(x10), "population": 256, "dims": 9, "sigma": 1.0}
            for tag, kind, fs in (("huber_f10", L.LOSS_HUBER, 10.0), ("mean_distance", L.LOSS_MEAN_DIST, 0.0)):
                g10, st10 = cma_loop(L, CMA, p10, b10, syn.TARGETS_D9, bounds_to_array, 256, kind, fs)
                k_g = args.c3_steps                        # 100 generations
                wc, _ = timed(ctl, L, g10, k_g, 3)
                ek, _ = p10.eval_population_timing()
                variant = p10.eval_population_info()[0]
                flop = EVAL_FLOPS_BY_VARIANT[(variant, "huber" if kind == L.LOSS_HUBER else "mean")]
                c3[tag] = {"iters_per_s": k_g / wc, "ms_per_iter": wc / k_g * 1e3, "generations_timed": k_g, "kernel_ms": ek,
                           "point_candidate_evals_per_s": len(x10) * 256 * k_g / wc,
                           "host_ms_per_generation": {"ask": st10["t_ask"] / st10["n"] * 1e3, "tell": st10["t_tell"] / st10["n"] * 1e3},
                           "kernel_variant": variant, "flop_per_eval": flop,
                           "valu_frac": len(x10) * 256 * flop / (ek / 1e3) / VALU_PEAK}
            # the same population through the GENERAL arithmetic (what rounds 1-5 measured at this shape; a D = 9 population around a
            # camera WITH a lens takes it): forced for ten generations
            os.environ["ALP_POP_NO_LENS_FREE"] = "1"
            try:
                g10, _ = cma_loop(L, CMA, p10, b10, syn.TARGETS_D9, bounds_to_array, 256, L.LOSS_HUBER, 10.0)
                wc, _ = timed(ctl, L, g10, 20, 3)
                ek, _ = p10.eval_population_timing()
                c3["huber_f10_general_arithmetic"] = {"iters_per_s": 20 / wc, "kernel_ms": ek, "kernel_variant": p10.eval_population_info()[0],
                                                      "flop_per_eval": EVAL_FLOPS, "valu_frac": len(x10) * 256 * EVAL_FLOPS / (ek / 1e3) / VALU_PEAK}
            finally:
                del os.environ["ALP_POP_NO_LENS_FREE"]
        out["c2_c3_10m"] = {"c2_projection": c2, "c3_cma": c3}
        out["roofline"].update({"c3_iters_per_s": c3["huber_f10"]["iters_per_s"], "c3_kernel_ms": c3["huber_f10"]["kernel_ms"],
                                "c3_evals_per_s": c3["huber_f10"]["point_candidate_evals_per_s"], "c3_valu_frac": c3["huber_f10"]["valu_frac"],
                                "c3_generations_timed": args.c3_steps, "c3_mean_distance_iters_per_s": c3["mean_distance"]["iters_per_s"],
                                "c3_kernel_variant": c3["huber_f10"]["kernel_variant"], "c3_flop_per_eval": c3["huber_f10"]["flop_per_eval"],
                                "c3_general_arithmetic_kernel_ms": c3["huber_f10_general_arithmetic"]["kernel_ms"],
                                "c3_general_arithmetic_iters_per_s": c3["huber_f10_general_arithmetic"]["iters_per_s"],
                                "c3_general_arithmetic_valu_frac": c3["huber_f10_general_arithmetic"]["valu_frac"]})
        del s10, x10, o10

    # ---------------------------------------------------------------- depth raster (1 GPU)
    if ctl.world == 1 and not args.no_raster:
        cam = syn.base_params(n_side)              # camera on the surface's west edge
        pv_cam = L.params_vector(cam)
        k_r = min(args.steps, 10)
        n_tri = 2 * (n_side - 1) ** 2
        W, H = int(cam["w"]), int(cam["h"])
        out["raster"] = {"frame": f"{W}x{H}", "vertices": n_total, "triangles": n_tri,
                         "call": "persp_proj(vert, vert, ind, params, offsets) as used by reverse_proj"}
        variants = [("implicit_grid", False)]
        if not args.no_raster_explicit:
            variants.append(("int32_indices", True))
            variants.append(("nodata_filtered_indices", True))
        for name, explicit in variants:
            ind = None
            if name == "int32_indices":     # time the index-array kernel itself (a full regular grid would be recognised)
                os.environ["ALP_NO_GRID_DETECT"] = "1"
                ind = syn.grid_indices(n_side, np.int32)
            elif explicit:   # what get_colored_surface returns for a DSM with nodata (surface.py:203-205): 0.5 % of the
                ind = syn.grid_indices(n_side, np.int32)   # vertices in 4 x 4 patches, their triangles removed
                hole = np.zeros((n_side, n_side), dtype=bool)
                rr = np.random.default_rng(5).integers(0, n_side - 4, (n_total // 3200, 2))
                for dy in range(4):
                    for dx in range(4):
                        hole[rr[:, 0] + dy, rr[:, 1] + dx] = True
                hole = hole.ravel()
                keep = ~(hole[ind[:, 0]] | hole[ind[:, 1]] | hole[ind[:, 2]])
                ind = ind[keep]
                del hole, keep
            t_mesh = time.perf_counter()
            mesh = L.Mesh(surf["vert"], None, ind, grid=None if explicit else (n_side, n_side))
            mesh.render_enqueue(pv_cam, surf["offsets"])
            L.synchronize()
            t_mesh = time.perf_counter() - t_mesh        # upload of the mesh + the first frame (incl. one-off tile bounds)
            os.environ.pop("ALP_NO_GRID_DETECT", None)
            # every timed frame is DRAWN: the library would serve a repeated view from the previous frame's
            # visibility buffer (reported separately as `same_view_again_ms`)
            os.environ["ALP_NO_VIS_CACHE"] = "1"
            wall_r, dev_r = timed(ctl, L, lambda: mesh.render_enqueue(pv_cam, surf["offsets"]), k_r, 2)
            os.environ.pop("ALP_NO_VIS_CACHE", None)
            full_before = mesh.frame_counts()
            wall_c, dev_c = timed(ctl, L, lambda: mesh.render_enqueue(pv_cam, surf["offsets"]), k_r, 1)
            assert mesh.frame_counts() == (full_before[0], full_before[1] + k_r + 1)
            img = mesh.fetch()
            # algorithmic bytes per frame: vertices 12 B (value == vert), indices 12 B per
            # triangle when explicit, visibility 8 B written + 8 B read per pixel, 12 B per pixel out
            n_tri_v = n_tri if ind is None else len(ind)
            alg = n_total * 12 + (n_tri_v * 12 if name == "int32_indices" else 0) + W * H * (16 + 12)
            tr, tr_src = pmc_traffic("raster_" + name)
            out["raster"][name] = {
                "ms_per_frame": wall_r / k_r * 1e3, "device_ms_per_frame": dev_r / k_r,
                "gvertices_per_s": n_total / (wall_r / k_r) / 1e9, "frames_timed": k_r, "triangles": n_tri_v,
                "covered_fraction": float((img[:, :, 0] > 0).mean()),
                "upload_inclusive_ms_first_frame": t_mesh * 1e3,
                # the same view again (sim_image then reverse_proj, example.py:28,31): resolve stage alone
                "same_view_again_ms": wall_c / k_r * 1e3, "same_view_again_device_ms": dev_c / k_r,
                "roofline": {"bound": "hbm", "achieved": alg / (dev_r / k_r / 1e3) / 1e9, "peak": HBM_PEAK / 1e9,
                             "unit": "GB/s", "frac": alg / (dev_r / k_r / 1e3) / HBM_PEAK,
                             "algorithmic_bytes_per_frame": alg,
                             "traffic": tr.get("hbm_bytes_per_frame") if tr else None, "traffic_source": tr_src,
                             # SURVEY 8(d) also counts the reference's separate remap pass (12 B read +
                             # 12 B written per pixel), which is fused away here
                             "frac_with_survey_remap_bytes": (alg + W * H * 24) / (dev_r / k_r / 1e3) / HBM_PEAK},
            }
            if not explicit:
                out["roofline"].update({"raster_ms_per_frame": dev_r / k_r, "raster_hbm_frac": alg / (dev_r / k_r / 1e3) / HBM_PEAK,
                                        "raster_frames_timed": k_r, "raster_same_view_again_ms": dev_c / k_r})
                # the frame is NOT HBM-bound (traffic = 0.93 x algorithmic at a quarter of the peak): its floor is set by the
                # L2 atomic line-requests and the vector instructions its kernels issue (counters of the committed probe of the
                # same workload, tools/raster_binding_roof.py) -- the fraction of THAT floor is the honest one
                br, br_src = committed_summary("raster_binding_roof")
                if br:
                    out["raster"][name]["binding_roof"] = {
                        "binding": br["binding"], "floor_ms": br["floor_ms"], "atomic_floor_ms": br["atomic_floor_ms"],
                        "valu_floor_ms": br["valu_floor_ms"], "frac": br["floor_ms"] / (dev_r / k_r),
                        "atomic_line_requests_per_frame": br["atomic_line_requests_per_frame"], "atomic_rate_per_s": br["atomic_rate_per_s"],
                        "valu_active_quad_cycles_per_frame": br["valu_active_quad_cycles_per_frame"], "source": br_src}
                    out["roofline"].update({"raster_binding_roof_frac": br["floor_ms"] / (dev_r / k_r), "raster_binding_roof": br["binding"],
                                            "raster_binding_roof_source": br_src})
            elif name == "int32_indices":
                out["roofline"].update({"raster_int32_indices_ms_per_frame": dev_r / k_r,
                                        "raster_int32_indices_hbm_frac": alg / (dev_r / k_r / 1e3) / HBM_PEAK})
            if not explicit:     # also with the distorted ground-truth pose (remap stage on)
                pv_dist = L.params_vector(syn.truth_params(n_side))
                os.environ["ALP_NO_VIS_CACHE"] = "1"
                wall_d, dev_d = timed(ctl, L, lambda: mesh.render_enqueue(pv_dist, surf["offsets"]), k_r, 2)
                os.environ.pop("ALP_NO_VIS_CACHE", None)
                out["raster"][name]["distorted_pose_ms_per_frame"] = wall_d / k_r * 1e3
            mesh.close()
            del img, ind

    # ---------------------------------------------------------------- the reference-typed call pair (1 GPU)
    df_full = ras_dev = None
    if ctl.world == 1 and not args.no_dropin and not args.no_raster:
        out["dropin_call"], df_full, _sim = dropin_leg(L, syn, surf, n_side, syn.base_params(n_side))
        # f4: set_gcp's gather of 1 127 GCPs from the resident coordinate image (gcp.py:644-648)
        from alproj_amd import project as aproj
        with aproj.reverse_proj_device(surf["vert"], None, syn.base_params(n_side), surf["offsets"], grid_shape=(n_side, n_side)) as rp:
            pick = df_full.iloc[:: max(1, len(df_full) // 1127)][:1127]
            uu, vv = pick["u"].to_numpy(), pick["v"].to_numpy()
            rp.lookup(uu, vv)
            L.kernel_timing(True)
            t_g = best_of(lambda: rp.lookup(uu, vv), 5)
            k_ms, k_n = L.kernel_time_ms()
            L.kernel_timing(False)
            t_join = None
            if len(df_full) < 30_000_000:
                import pandas as pd
                m_df = pd.DataFrame({"u_org": uu, "v_org": vv, "u_sim": uu, "v_sim": vv})
                t_join = best_of(lambda: pd.merge(m_df, df_full, how="left", left_on=["u_sim", "v_sim"], right_on=["u", "v"]), 1)
            out["f4_set_gcp_gather"] = {"gcps": len(uu), "call_ms": t_g * 1e3, "kernel_ms": k_ms / max(k_n, 1),
                                        "reference_merge_with_the_table_ms_same_box": None if t_join is None else t_join * 1e3,
                                        "note": "launch-latency bound: 1127 x 12 B gathered"}
            # f2 fed from the RESIDENT frame: reverse_proj -> to_geotiff's compute without the table in between
            # (ReverseProjection.rasterize: alp_render_rasterize_plan + alp_render_rasterize); the 63 MB image goes up, the raster comes back
            rp.rasterize(_sim, ["B", "G", "R"], 1.0, ["B", "G", "R"], True, 1.0, "mean")      # warm-up
            L.kernel_timing(True)
            walls = []
            for _ in range(4):             # the call allocates and frees 5.3 GB of HBM and a 237 MB host array each time: its wall time is
                L.kernel_time_ms()         # 15-27 ms, with an occasional 270-430 ms call (allocation, not kernels: kernel_ms does not move)
                t = time.perf_counter()
                ras_dev, _bounds = rp.rasterize(_sim, ["B", "G", "R"], 1.0, ["B", "G", "R"], True, 1.0, "mean")
                walls.append(time.perf_counter() - t)
                k_ms, k_n = L.kernel_time_ms()
            L.kernel_timing(False)
            out["f2_device_fed"] = {"call": "ReverseProjection.rasterize(image, ...) = reverse_proj + to_geotiff compute, no table",
                                    "points": len(df_full), "raster": list(ras_dev.shape),
                                    "call_ms_incl_image_upload_and_raster_fetch": {"best": min(walls) * 1e3, "median": float(np.median(walls)) * 1e3,
                                                                                   "worst": max(walls) * 1e3, "calls": len(walls)},
                                    "kernel_ms": k_ms, "kernel_sections": k_n}
        aproj.clear_mesh_cache()
    if ctl.world == 1 and not args.no_next_rows:
        out["next_rows"] = next_rows_leg(L, syn, orc, df_full, ras_dev)
        if "f2_device_fed" in out and "f2_rasterize_points" in out["next_rows"]:
            out["f2_device_fed"]["byte_identical_to_the_table_path"] = out["next_rows"]["f2_rasterize_points"].pop("equals_device_fed_raster")
        del df_full
        # the whole pipeline of examples/pipeline_synthetic.py (= the reference's example.py) at the reference's sizes:
        # 5616 x 3744 image, 6000 x 6000 = 36 M-vertex surface (distance 3000 m at 1 m, example.py:22,25)
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import pipeline_synthetic as pipe
        t_p = time.perf_counter()
        r = pipe.run(n=6000, w=5616, h=3744, generations=150, verbose=False)
        out["pipeline_full_size"] = {"vertices": 36_000_000, "image": "5616x3744", "total_s": time.perf_counter() - t_p,
                                     "stage_ms": {k: v * 1e3 for k, v in r["times"].items()}, "gcps": list(r["gcps"]),
                                     "reproj_px_initial": r["reproj_px_initial"], "reproj_px_final": r["reproj_px_final"],
                                     "georectified_rows": r["georectified_rows"], "raster_shape": list(r["raster_shape"])}

    # ---------------------------------------------------------------- GCP-scale optimiser (1 GPU)
    # the reference's own problem size: 1127 GCPs, pop 50, D = 9, 300 generations, Huber f = 10
    # (docs/usage.md:335, example.py:51-54); BASELINE.md measured 61.7 ms/generation for the
    # reference's inner loop (without the sampler) in the survey container
    if ctl.world == 1 and not args.no_cma:
        import pandas as pd
        from alproj_amd import optimize as aopt
        tp = syn.truth_params(316)
        gx = syn.gcp_points(1127, tp, seed=3)
        with L.Points(gx, [tp["x"], tp["y"], tp["z"]], "f64") as gp:
            gp.project(L.params_vector(tp))
            gu, gv = gp.fetch()
        guv = np.stack([gu, gv], 1) + np.random.default_rng(3).normal(0, 1.0, (1127, 2))
        init = dict(tp, pan=tp["pan"] + 2, tilt=tp["tilt"] - 1.5, fov=tp["fov"] + 3, x=tp["x"] + 4)
        o = aopt.CMAOptimizer(pd.DataFrame(gx, columns=["x", "y", "z"]), pd.DataFrame(guv, columns=["u", "v"]), init)
        o.set_target(syn.TARGETS_D9)
        t0 = time.perf_counter()
        # the DEFAULT call: no precision named -> the reference's float64 at this size (alproj_amd.optimize.default_precision)
        _, err = o.optimize(generation=300, sigma=1.0, population_size=50, f_scale=10.0, seed=7, progress=False)
        dt = time.perf_counter() - t0
        out["cma_gcp_scale"] = {"gcps": 1127, "population": 50, "dims": 9, "generations": 300,
                                "precision": aopt.default_precision(1127) + " (the default: nothing passed)",
                                "ms_per_generation": dt / 300 * 1e3, "generations_per_s": 300 / dt,
                                "final_mean_distance_px": err,
                                "reference_ms_per_generation_survey_container": 61.7}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1)
    if ctl.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(orc, truth, base, targets, bounds_to_array, xyz_l, obs, n_total, args.pop)
        out["speedup_vs_cpu_baseline"] = gpts / out["cpu_baseline"]["value"]

    if ctl.world > 1 and ctl.rank == 0:
        exp = expectation_from_one_gpu(ctl.world, out.get("cma", {}).get("all_reduce_ms"), n_total, args.pop if "cma" in out else None)
        if exp:
            exp["projection_measured_over_expected"] = (out["roofline"]["kernel_ms"] / exp["projection_kernel_ms"]) if exp.get("projection_kernel_ms") else None
            if "cma" in out and exp.get("cma_ms_per_iter"):
                exp["cma_kernel_measured_over_expected"] = out["cma"]["roofline"]["kernel_ms"] / exp["cma_kernel_ms"]
                exp["cma_ms_per_iter_measured_over_expected"] = out["cma"]["ms_per_iter"] / exp["cma_ms_per_iter"]
            out["expected_from_1gpu"] = exp
    pts.close()
    # roofline.traffic, live: after everything else (the GPU is idle, the big host arrays are still this process's own)
    if ctl.world == 1 and args.precision == "f32" and not args.no_live_traffic and not args.debug_share_device:
        t_pmc = time.perf_counter()
        measured, why = live_traffic(n_total)
        out["roofline"]["traffic_committed_summary"] = out["roofline"].get("traffic")
        if measured is not None:
            out["roofline"]["traffic"] = measured
            out["roofline"]["traffic_source"] = why
        else:
            out["roofline"]["traffic_live_measurement_failed"] = why
            out["roofline"]["traffic_source"] = (f"{out['roofline'].get('traffic_source') or 'no committed summary'}; NOT measured in this run: {why}")
        out["roofline"]["traffic_live_seconds"] = time.perf_counter() - t_pmc
    out["roofline"], out["roofline_detail"] = driver_roofline(out["roofline"])
    ctl.barrier()            # every rank has finished its collectives
    adist.shutdown()         # ncclCommDestroy (no-op without a communicator)
    ctl.close()
    sys.stdout.flush()
    if ctl.rank == 0:
        emit(json.dumps(out))


if __name__ == "__main__":
    main()
