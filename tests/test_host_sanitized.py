"""The HIP-free host side of libalproj_hip.so (alproj_amd/csrc/host/: error state, fold_pose, the threads behind
alp_host_hash64 / alp_host_minmax / alp_host_prefault, the grid-recognition threads of alp_mesh_create, the conversion
workers of alp_projected_fetch, the argmin / confirmation-band selection of alp_eval_population_wait) compiled WITHOUT HIP
and run under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer, each in a child process.

The driver (csrc/host/alp_host_selfcheck.cpp) calls every helper with sizes that straddle its thread thresholds, with 1 to
64 threads, from several caller threads at once, and compares with serial restatements.  The sanitizers are shown to be
awake first: each build must REPORT the deliberate defect made for it (--canary).  ThreadSanitizer builds use g++ only:
the ROCm clang's ThreadSanitizer reported the canary race in some settings (10 of 10 runs as a child of Python) and
missed it in others (0 of 10 and 1 of 5 runs from a shell), g++'s reported it in every run of both -- so only g++'s
silence is taken as a verdict; AddressSanitizer / UBSan run under both compilers (profiles/r06_host_sanitizers.txt).

Also here: the recycled result memory of alproj_amd/_lib.py (weakref finalizers + a lock) under a stress loop from
several threads with the collector forced."""
import gc
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alproj_amd import _build  # noqa: E402

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1")
REPORT_WORDS = ("Sanitizer", "runtime error", "CHECK FAILED", "data race")

BUILDS = [("plain", "clang"), ("asan", "clang"), ("asan", "gcc"), ("tsan", "gcc")]


def _exe(kind, compiler):
    if _build.host_compiler(compiler) is None:
        pytest.skip(f"no {compiler} compiler")
    return _build.build_host(kind, compiler)


def _run(exe, *args, timeout=900):
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=timeout, env=ENV)


@pytest.mark.parametrize("kind,compiler,defect,word", [
    ("asan", "clang", "overflow", "AddressSanitizer: heap-buffer-overflow"),
    ("asan", "gcc", "overflow", "AddressSanitizer: heap-buffer-overflow"),
    ("asan", "clang", "shift", "runtime error: shift exponent 64"),
    ("asan", "gcc", "shift", "runtime error: shift exponent 64"),
    ("tsan", "gcc", "race", "ThreadSanitizer: data race"),
])
def test_the_sanitizer_is_awake(kind, compiler, defect, word):
    """a build whose sanitizer misses the defect planted for it would also miss a real one"""
    r = _run(_exe(kind, compiler), "--canary", defect)
    assert r.returncode != 0 and word in r.stderr, (r.returncode, r.stderr[-2000:])


@pytest.mark.parametrize("kind,compiler", BUILDS)
@pytest.mark.parametrize("mode", ["concurrent", "serial"])
def test_host_selfcheck_is_clean(kind, compiler, mode):
    r = _run(_exe(kind, compiler), *([] if mode == "concurrent" else ["--serial"]))
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-4000:])
    assert "host selfcheck ok" in r.stdout
    assert not any(w in r.stderr for w in REPORT_WORDS), r.stderr[-4000:]


def test_library_and_sanitized_build_share_their_sources():
    """what runs under the sanitizers is what ships: the library's source list names host/alp_host.cpp, and neither it nor
    the header it shares with the .hip units includes a HIP header"""
    assert "host/alp_host.cpp" in _build.SOURCES
    for name in ("alp_host.h", "alp_host.cpp"):
        text = open(os.path.join(_build.HOST_DIR, name)).read()
        assert "hip_runtime" not in text and "#include <hip" not in text, name
    # the moved code is gone from the HIP units: one definition each
    for unit, words in (("alp_core.hip", ("hash_slice", "MADV_POPULATE_WRITE", "void fold_pose(")),
                        ("alp_points.hip", ("void convert_slice(", "std::thread")),
                        ("alp_raster.hip", ("struct HostGridCheck", "void grid_rows_check("))):
        text = open(os.path.join(_build.CSRC, unit)).read()
        for w in words:
            assert w not in text, (unit, w)


# ---------------------------------------------------------------- the recycled result memory of _lib.py
def test_result_pool_under_threads_and_forced_collection():
    from alproj_amd import _lib as L
    old_cap = L._pool_cap
    L.clear_result_pool()
    L.set_result_pool(5 * L._POOL_MIN)           # room for a few buffers: every release past that evicts
    sizes = [L._POOL_MIN, L._POOL_MIN + 4096, 2 * L._POOL_MIN, 3 * L._POOL_MIN]
    errors = []
    stop = threading.Event()

    def check_invariants():
        with L._pool_lock:
            held = sum(nb * len(bufs) for nb, bufs in L._pool.items())
            assert held == L._pool_bytes, (held, L._pool_bytes)
            assert L._pool_bytes <= L._pool_cap
            assert sorted((nb, id(b)) for nb, bufs in L._pool.items() for b in bufs) == sorted(L._pool_age)
            assert all(len(bufs) <= L._POOL_PER_SIZE and bufs for bufs in L._pool.values())

    def worker(seed):
        rng = np.random.default_rng(seed)
        try:
            held = []
            for k in range(60):
                nb = sizes[int(rng.integers(len(sizes)))]
                a = L.result_empty((nb // 4,), np.float32)
                assert a.nbytes == nb and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]
                a[:7] = seed                      # the memory is this thread's alone: nobody else may hold the same buffer
                a[-7:] = seed
                v = a[3:11]                       # a view keeps the buffer alive after the array is dropped
                held.append((a if k % 3 else v, seed))
                if len(held) > 3:
                    b, s = held.pop(int(rng.integers(len(held))))
                    assert float(b[0]) == s           # still this thread's bytes
                    del b
                if k % 5 == 0:
                    gc.collect()                  # finalizers run here, possibly inside another thread's locked region
                if k % 17 == 0:
                    check_invariants()
            for b, s in held:
                assert float(b[0]) == s
        except Exception as e:                    # noqa: BLE001
            errors.append(repr(e))

    def collector():
        while not stop.wait(0.002):
            gc.collect()

    def resizer():
        k = 0
        while not stop.wait(0.003):
            L.set_result_pool((3 + k % 4) * L._POOL_MIN)
            if k % 7 == 0:
                L.clear_result_pool()
            k += 1

    try:
        threads = [threading.Thread(target=worker, args=(s,)) for s in range(1, 5)]
        helpers = [threading.Thread(target=collector), threading.Thread(target=resizer)]
        for t in threads + helpers:
            t.start()
        for t in threads:
            t.join()
        stop.set()
        for t in helpers:
            t.join()
        assert not errors, errors
        gc.collect()
        check_invariants()
        assert L.POOL_STATS["hits"] > 0 and L.POOL_STATS["evicted"] > 0
    finally:
        L.clear_result_pool()
        L.set_result_pool(old_cap)
