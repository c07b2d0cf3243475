"""The line bench.py prints, as the driver stores it: `roofline` must carry the whole metric within the 24 keys the
driver keeps (task 1).  No GPU: the split is exercised on the committed N = 1 line and on a canned one."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MUST_CARRY = ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
              "scaled_1e-5_pass", "strict_1e-5_relative_pass", "cma_iters_per_s", "cma_kernel_ms", "cma_valu_frac", "cma_all_reduce_ms",
              "f64_gpoints_per_s", "f64_hbm_frac", "f64_strict_1e-5_pass", "c3_iters_per_s", "c3_kernel_ms", "c3_valu_frac",
              "raster_ms_per_frame", "raster_hbm_frac", "raster_binding_roof_frac", "raster_int32_indices_ms_per_frame"]


def canned():
    """what the legs of a default N = 1 run put into the flat dict before the split (names as in bench.main)"""
    flat = {k: 1.0 for k in MUST_CARRY}
    flat.update({"bound": "hbm", "unit": "GB/s", "kernel": "project_kernel",
                 "traffic_source": "profiles/...", "kernel_ms_median_of_single_launches": 0.3, "single_launches": 24,
                 "vertices_per_launch": 100_000_000, "cma_generations_timed": 50, "cma_population": 2048, "cma_dims": 21,
                 "cma_valu_frac_at_survey_100_flop": 0.6, "cma_all_reduce_share": 0.0, "f64_kernel_ms": 0.6, "f64_max_err_rel": 1e-16,
                 "f64_cma_iters_per_s": 1.5, "c2_gpoints_per_s_kernel_median": 250.0, "c2_hbm_frac_at_median": 0.6,
                 "c3_evals_per_s": 8e11, "c3_generations_timed": 100, "c3_mean_distance_iters_per_s": 340.0,
                 "raster_frames_timed": 10, "raster_same_view_again_ms": 0.14, "raster_binding_roof": "l2_atomics",
                 "raster_binding_roof_source": "profiles/...", "raster_int32_indices_hbm_frac": 0.28})
    return flat


def test_roofline_fits_the_drivers_cap_and_carries_all_five_configs():
    assert list(bench.ROOFLINE_KEYS) == MUST_CARRY and len(bench.ROOFLINE_KEYS) <= bench.ROOFLINE_CAP == 24
    flat = canned()
    roof, detail = bench.driver_roofline(flat)
    assert list(roof) == MUST_CARRY                       # order too: the driver keeps the FIRST 24
    assert len(roof) <= 24 and all(roof[k] is not None for k in MUST_CARRY)
    assert not set(roof) & set(detail) and set(roof) | set(detail) == set(flat)
    # a leg that did not run (N > 1: no float64 / 10 M / render legs) leaves None, never a missing or shifted key
    part = {k: v for k, v in flat.items() if not k.startswith(("f64_", "c3_", "raster_"))}
    roof2, _ = bench.driver_roofline(part)
    assert list(roof2) == MUST_CARRY and roof2["c3_iters_per_s"] is None and roof2["frac"] == 1.0
    json.dumps(roof)


def test_committed_line_of_this_round_obeys_the_cap():
    """the round's own committed N = 1 line (profiles/r06_bench_default_run.json, once it exists) was split by the same rule;
    round 5's line holds round 5's names (bytes_per_vertex where scaled_1e-5_pass now is: it moved to `config`)"""
    f = os.path.join(ROOT, "profiles", "r06_bench_default_run.json")
    if os.path.exists(f):
        line = json.load(open(f))
        assert list(line["roofline"]) == MUST_CARRY
        missing = [k for k in MUST_CARRY if line["roofline"][k] is None]
        assert not missing, missing
        assert 0 < line["roofline"]["raster_binding_roof_frac"] <= 1.05 and line["roofline"]["raster_ms_per_frame"] > 0
        # which number meets which tolerance, readable from the driver's record alone (task 2)
        assert line["roofline"]["scaled_1e-5_pass"] == 1.0 and line["roofline"]["f64_strict_1e-5_pass"] == 1.0
        assert "1e-5|ref|" in line["config"]["tol_strict"] and "max(|ref|,w)" in line["config"]["tol_scaled"]
        assert all(len(v) <= 128 for v in line["config"].values() if isinstance(v, str))
    f5 = os.path.join(ROOT, "profiles", "r05_bench_default_run.json")
    if os.path.exists(f5):
        keys5 = [("bytes_per_vertex" if k == "scaled_1e-5_pass" else k) for k in MUST_CARRY]
        assert list(json.load(open(f5))["roofline"]) == keys5


def test_parity_report_states_both_readings_of_the_tolerance():
    """north_star: "1e-5 relative".  strict = of |ref|, scaled = of max(|ref|, image width): the report carries both
    definitions and both pass fractions and says which one a mode meets"""
    import numpy as np
    ref = np.array([[2000.0, 1000.0], [3.0, 4000.0], [0.5, 10.0]])
    exact = bench.parity_report(ref.copy(), ref, 5616.0)
    assert exact["meets"] == "strict" and exact["strict_1e-5_relative_pass_fraction"] == 1.0 == exact["scaled_1e-5_pass_fraction"]
    f32ish = bench.parity_report(ref + 2e-4, ref, 5616.0)            # the float32 floor: 2e-4 px on every value
    assert f32ish["meets"] == "scaled" and f32ish["scaled_1e-5_pass_fraction"] == 1.0
    assert f32ish["strict_1e-5_relative_pass_fraction"] == pytest.approx(3 / 6)       # |ref| >= 20 px pass, 3, 0.5 and 10 do not
    bad = bench.parity_report(ref + 1.0, ref, 5616.0)
    assert bad["meets"] == "neither"
    assert set(exact["definitions"]) == {"strict", "scaled"} and "|ref|" in exact["definitions"]["strict"]


def test_binding_roof_summary_is_consistent():
    br, src = bench.committed_summary("raster_binding_roof")
    assert br is not None and src.startswith("profiles/")
    assert br["floor_ms"] == max(br["atomic_floor_ms"], br["valu_floor_ms"])
    assert br["atomic_floor_ms"] == pytest.approx(br["atomic_line_requests_per_frame"] / br["atomic_rate_per_s"] * 1e3)
    assert br["valu_floor_ms"] == pytest.approx(br["valu_active_quad_cycles_per_frame"] * 4 / (br["simds"] * br["clock_hz"]) * 1e3)
    assert br["binding"] in ("l2_atomics", "valu")


def test_expectation_for_n_gpus_is_kernel_time_over_n(tmp_path, monkeypatch):
    one = {"n_gpus": 1, "config": {"vertices": 100_000_000}, "roofline": {"kernel_ms": 0.32, "cma_kernel_ms": 214.0},
           "cma": {"ms_per_iter": 216.0}}
    (tmp_path / "r05_bench_default_run.json").write_text(json.dumps(one))
    monkeypatch.setattr(bench, "PROFILES", str(tmp_path))
    e = bench.expectation_from_one_gpu(8, 0.05)
    assert e["projection_kernel_ms"] == pytest.approx(0.04) and e["projection_gpoints_per_s"] == pytest.approx(2500.0)
    assert e["cma_kernel_ms"] == pytest.approx(26.75) and e["cma_ms_per_iter"] == pytest.approx(26.75 + 0.05 + 2.0)
    assert bench.expectation_from_one_gpu(8, 0.05, vertices=100_000_000)["cma_kernel_ms"] == pytest.approx(26.75)
    assert bench.expectation_from_one_gpu(8, 0.05, vertices=4_000_000) is None          # another workload: no expectation
    monkeypatch.setattr(bench, "PROFILES", str(tmp_path / "nothing"))
    assert bench.expectation_from_one_gpu(8, 0.05) is None


def _tree_listing(root):
    out = set()
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in (".git", "__pycache__", ".pytest_cache")]
        out.update(os.path.join(d, f) for f in files)
        out.update(os.path.join(d, x) for x in dirs)
    return out


def test_live_traffic_never_takes_the_line_down(monkeypatch):
    """bench.live_traffic: without rocprofv3, or when its passes fail (this container has no GPU), the answer is (None, why)
    and bench.py keeps the committed summary's figure"""
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    got, why = bench.live_traffic(1_000_000)
    assert got is None and "rocprofv3" in why
    monkeypatch.undo()
    if shutil.which("rocprofv3"):
        got, why = bench.live_traffic(100_000, timeout_s=120.0)
        assert why and (got is None or got > 0)                 # no GPU here: the passes fail and the reason comes back (on a GPU box: a figure)


def test_live_traffic_is_skipped_under_a_profiler():
    """`rocprofv3 ... -- python3 bench.py` must not start a second profiler inside the first (round-5 advisor)"""
    assert not bench.under_a_profiler({"PATH": "/usr/bin"})
    assert bench.under_a_profiler({"ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.under_a_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so.1"})
    assert bench.under_a_profiler({"ROCPROFILER_PC_SAMPLING_BETA_ENABLED": "1"})
    old = os.environ.get("ROCP_TOOL_LIBRARIES")
    os.environ["ROCP_TOOL_LIBRARIES"] = "x"
    try:
        got, why = bench.live_traffic(1000)
    finally:
        if old is None:
            del os.environ["ROCP_TOOL_LIBRARIES"]
        else:
            os.environ["ROCP_TOOL_LIBRARIES"] = old
    assert got is None and "profiler" in why


@pytest.mark.parametrize("stub", ["works", "fails", "hangs"])
def test_live_traffic_needs_no_writable_checkout_and_leaves_nothing_behind(tmp_path, monkeypatch, stub):
    """A stub `rocprofv3` in front of PATH (writes the results database the tool reads; or exits 1; or sleeps past the limit)
    and a READ-ONLY checkout: the live pass works entirely in a temporary directory outside the tree, the tree is untouched
    byte for byte, the temporary directory is gone afterwards on every way out, and a failure names its reason (the caller
    writes it to roofline_detail.traffic_live_measurement_failed and keeps the committed figure)."""
    import stat
    import tempfile
    bindir = tmp_path / "bin"
    bindir.mkdir()
    script = bindir / "rocprofv3"
    script.write_text("""#!/usr/bin/env python3
import os, sqlite3, sys, time
mode = %r
if mode == "fails":
    sys.exit(1)
if mode == "hangs":
    time.sleep(600)
a = sys.argv[1:]
d, o, counter = a[a.index("-d") + 1], a[a.index("-o") + 1], a[a.index("--pmc") + 1]
assert not os.path.abspath(d).startswith(%r), d            # never inside the checkout
db = sqlite3.connect(os.path.join(d, o + "_results.db"))
db.execute("create table counters_collection (kernel_name text, counter_name text, value real, dispatch_id integer)")
kib = {"FETCH_SIZE": 1200000000 / 2 / 1024, "WRITE_SIZE": 800000000 / 1024}[counter]      # per launch; FETCH is doubled by the tool
for i in range(3):
    db.execute("insert into counters_collection values (?, ?, ?, ?)",
               ("void alp::project_kernel<float>(float const*, float const*)", counter, kib, i))
db.commit()
""" % (stub, ROOT))
    script.chmod(0o755)
    monkeypatch.setenv("PATH", f"{bindir}{os.pathsep}{os.environ['PATH']}")
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    monkeypatch.setenv("TMPDIR", str(scratch))
    monkeypatch.setattr(tempfile, "tempdir", None)              # re-read TMPDIR
    # a read-only checkout, as far as this process can tell: every directory bench.py could be tempted to write into
    before = _tree_listing(ROOT)
    locked = []
    for d in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "profiles")):
        mode = stat.S_IMODE(os.stat(d).st_mode)
        os.chmod(d, mode & ~0o222)
        locked.append((d, mode))
    try:
        got, why = bench.live_traffic(100_000_000, timeout_s=3.0 if stub == "hangs" else 60.0)
    finally:
        for d, mode in locked:
            os.chmod(d, mode)
        monkeypatch.setattr(tempfile, "tempdir", None)
    if stub == "works":
        assert got == pytest.approx(2.0e9) and "measured in this run" in why
    elif stub == "fails":
        assert got is None and "exited with" in why
    else:
        assert got is None and "did not finish within" in why
    assert _tree_listing(ROOT) == before                         # nothing created in (or removed from) the checkout
    assert os.listdir(scratch) == []                             # and the temporary directory is gone
