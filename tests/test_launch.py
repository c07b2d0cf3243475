"""The torch-free launcher and control plane (alproj_amd/launch.py) that `python bench.py --gpus N` uses to start its own
ranks: CPU tests with bench.py's stub worker (--launch-selftest: rendezvous of a 128-byte id, barriers, max over
ranks, gather -- what a rank does around the real benchmark, without GPU or library)."""
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from alproj_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=120, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", launch.HUB_ENV, launch.KEY_ENV, launch.KFD_ENV,
              "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _pids(stderr):
    return [int(m) for m in re.findall(r"selftest: rank \d+ pid (\d+)", stderr)]


def _gone(pids, within_s=10.0):
    t0 = time.time()
    while time.time() - t0 < within_s:
        alive = []
        for p in pids:
            try:
                os.kill(p, 0)
                alive.append(p)
            except ProcessLookupError:
                pass
        if not alive:
            return True
        time.sleep(0.1)
    return False


@pytest.mark.parametrize("n", [2, 3])
def test_bench_launches_its_own_ranks(n):
    r = _run(["--gpus", str(n), "--launch-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                     # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["selftest"] and out["n_gpus"] == n and out["max_over_ranks"] == 10.0 + n - 1
    ranks = out["ranks"]
    assert [x["rank"] for x in ranks] == list(range(n)) and [x["local_rank"] for x in ranks] == list(range(n))
    assert all(x["world"] == n and x["hub"] == "parent" for x in ranks)
    assert all(x["env"] == {"RANK": str(i), "LOCAL_RANK": str(i), "WORLD_SIZE": str(n)} for i, x in enumerate(ranks))
    assert len({x["uid_sha256"] for x in ranks}) == 1          # rank 0's 128 bytes reached everybody
    assert len({x["pid"] for x in ranks}) == n and len({x["ppid"] for x in ranks}) == 1     # fresh children of one launcher
    assert sorted(_pids(r.stderr)) == sorted(x["pid"] for x in ranks) and _gone(_pids(r.stderr))


def test_a_failing_rank_ends_the_job_and_the_other_ranks():
    r = _run(["--gpus", "3", "--launch-selftest", "--selftest-fail-rank", "1"])
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert not r.stdout.strip()                                # no JSON line from a failed job
    assert "rank 1" in r.stderr and "exited with 7" in r.stderr
    pids = _pids(r.stderr)
    assert len(pids) == 3 and _gone(pids)


def test_one_wall_clock_timeout_for_the_whole_job():
    t0 = time.time()
    r = _run(["--gpus", "2", "--launch-selftest", "--selftest-hang-rank", "1", "--launch-timeout", "4"])
    assert r.returncode == 124 and time.time() - t0 < 30, (r.returncode, r.stderr[-2000:])
    pids = _pids(r.stderr)
    assert len(pids) == 2 and _gone(pids)


def test_gpus_1_needs_no_launcher_and_world_mismatch_is_refused():
    r = _run(["--gpus", "1", "--launch-selftest"])
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 1 and len(_pids(r.stderr)) == 1
    r = _run(["--gpus", "2", "--launch-selftest"], env={"WORLD_SIZE": "1", "RANK": "0"})      # a launcher's ranks, wrong --gpus
    assert r.returncode == 2


def test_ranks_of_a_foreign_launcher_meet_at_rank_0(tmp_path):
    """WORLD_SIZE already set (torch.distributed.run, or here: three plain children with its environment): rank 0 hosts
    the hub and announces it through a file keyed by MASTER_ADDR / MASTER_PORT; the file is gone afterwards"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    base = dict(os.environ, WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, TMPDIR=str(tmp_path))
    base.pop(launch.HUB_ENV, None)
    base.pop("XDG_RUNTIME_DIR", None)                          # the announcement goes to <TMPDIR>/alproj_<uid>/
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "3", "--launch-selftest"], cwd=ROOT,
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (2, 0, 1)]                               # rank 0 is not the first to start
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], [o[1][-800:] for o in outs]
    out = json.loads(outs[1][0])
    assert out["n_gpus"] == 3 and all(x["hub"] == "rank0" for x in out["ranks"]) and len({x["uid_sha256"] for x in out["ranks"]}) == 1
    assert not outs[0][0].strip() and not outs[2][0].strip()
    private = tmp_path / f"alproj_{os.getuid()}"
    assert private.is_dir() and (private.stat().st_mode & 0o777) == 0o700 and not os.listdir(private)     # the file is gone afterwards


def test_hub_announcement_is_private_and_its_key_is_not_derivable(tmp_path, monkeypatch):
    """the key of rank 0's hub is random and travels in a 0600 file inside a 0700 directory of this user;
    a directory or file somebody else could have prepared is refused."""
    import stat
    monkeypatch.delenv("XDG_RUNTIME_DIR", raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    monkeypatch.setattr(tempfile, "tempdir", None)             # re-read TMPDIR
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500")):
        monkeypatch.setenv(k, v)
    path = launch._rendezvous_file()
    d = os.path.dirname(path)
    assert d == str(tmp_path / f"alproj_{os.getuid()}") and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    keys = set()
    for _ in range(2):                                          # two jobs with the same public tag: two different keys
        hub = launch.Hub(1)
        launch._announce(path, hub.address, hub.authkey)
        assert stat.S_IMODE(os.lstat(path).st_mode) == 0o600
        addr, key = launch._read_announcement(path)
        assert addr == hub.address and key == hub.authkey and len(key) == 16
        keys.add(key)
        hub.close()
    assert len(keys) == 2
    # a world-readable file, a symlink, a loose directory: all refused
    os.chmod(path, 0o644)
    with pytest.raises(launch.LaunchError):
        launch._read_announcement(path)
    os.unlink(path)
    real = tmp_path / "elsewhere"
    real.write_text('{"host": "127.0.0.1", "port": 1, "key": "00"}')
    os.chmod(real, 0o600)
    os.symlink(real, path)
    with pytest.raises(OSError):
        launch._read_announcement(path)
    os.unlink(path)
    os.chmod(d, 0o755)
    with pytest.raises(launch.LaunchError):
        launch._rendezvous_file()
    os.chmod(d, 0o700)


def test_a_stranger_at_the_door_does_not_stop_the_hub():
    """a connection that fails the key handshake (another job's rank, a port scanner) is turned away and the hub keeps
    waiting for its own ranks"""
    from multiprocessing.connection import Client
    hub = launch.Hub(2).start()
    with pytest.raises(Exception):
        Client(hub.address, authkey=b"not the key of this hub")
    s = socket.create_connection(hub.address)          # ... and one that says nothing at all
    s.close()
    res = []

    def rank(r):
        c = launch.Control(r, 2, r, launch.Control._connect(lambda: (hub.address, hub.authkey), r, 30.0))
        res.append(c.gather(r))
        c.close()

    th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    hub.join(5)
    assert hub.error is None and res == [[0, 1], [0, 1]]


def test_a_silent_stranger_who_stays_connected_is_hung_up_on(monkeypatch):
    """a peer that connects and never answers the key challenge used to park the hub inside Listener.accept's blocking
    handshake for ever (the listening socket's timeout does not apply to the accepted connection): the ranks behind it in
    the queue never got in.  Now the handshake has its own deadline."""
    monkeypatch.setattr(launch, "HANDSHAKE_TIMEOUT_S", 0.5)
    hub = launch.Hub(2).start()
    silent = socket.create_connection(hub.address)     # first in the queue, stays open, sends nothing
    res = []

    def rank(r):
        c = launch.Control(r, 2, r, launch.Control._connect(lambda: (hub.address, hub.authkey), r, 30.0))
        res.append(c.gather(r))
        c.close()

    t0 = time.time()
    th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    hub.join(5)
    silent.close()
    assert hub.error is None and res == [[0, 1], [0, 1]]
    assert time.time() - t0 < 10


def test_nothing_received_is_unpickled():
    """a peer that knows the key and speaks pickle (Connection.send) gets an error; the hub does not build its object"""
    from multiprocessing.connection import Client
    hub = launch.Hub(1).start()

    class Bomb:
        def __reduce__(self):
            return (os.system, ("touch /tmp/alproj_pickle_bomb_%d" % os.getpid(),))

    c = Client(hub.address, authkey=hub.authkey)
    c.send(Bomb())
    hub.join(10)
    assert hub.error and "malformed" in hub.error
    assert not os.path.exists("/tmp/alproj_pickle_bomb_%d" % os.getpid())
    with pytest.raises(TypeError):
        launch._enc(object())
    assert launch._dec(json.loads(json.dumps(launch._enc({"id": b"\x00\xff", "t": (1, 2.5, None)})))) == {"id": b"\x00\xff", "t": [1, 2.5, None]}


def _fake_kfd(root, gpus, cpus=2):
    for i in range(cpus + gpus):
        d = root / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if i < cpus else 0}\nsimd_count {0 if i < cpus else 1024}\nmem_banks_count 1\n")
    return str(root)


def test_preflight_counts_gpus_from_sysfs_without_hip(tmp_path):
    """VERDICT round 4, task 3(a): fewer GPUs than --gpus -> one clear line, exit 2, nothing spawned"""
    eight = _fake_kfd(tmp_path / "eight", 8)
    four = _fake_kfd(tmp_path / "four", 4)
    assert launch.gpu_nodes(eight) == (8, eight) and launch.gpu_nodes(four)[0] == 4
    assert launch.gpu_nodes(str(tmp_path / "absent"))[0] is None
    r = _run(["--gpus", "8", "--launch-selftest"], env={launch.KFD_ENV: eight})
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 8 and len(_pids(r.stderr)) == 8
    r = _run(["--gpus", "8", "--launch-selftest"], env={launch.KFD_ENV: four})
    assert r.returncode == 2 and not r.stdout.strip() and not _pids(r.stderr)
    assert "--gpus 8 but this node offers 4" in r.stderr and "nothing was started" in r.stderr
    r = _run(["--gpus", "4", "--launch-selftest"], env={launch.KFD_ENV: eight, "HIP_VISIBLE_DEVICES": "0,1"})
    assert r.returncode == 2 and "offers 2" in r.stderr and "VISIBLE_DEVICES" in r.stderr
    r = _run(["--gpus", "2", "--launch-selftest"], env={launch.KFD_ENV: str(tmp_path / "absent")})
    assert r.returncode == 2 and "no GPU on this node" in r.stderr
    # the real benchmark (not the stub) in this GPU-less container: refused by the pre-flight before anything is built or started
    r = _run(["--gpus", "2"], env={launch.KFD_ENV: str(tmp_path / "absent")})
    assert r.returncode == 2 and "no GPU on this node" in r.stderr


def test_spawn_passes_the_operators_ipc_setting_on_and_says_when_it_sets_one(tmp_path):
    show = tmp_path / "show.py"
    show.write_text("import os, sys\nsys.path.insert(0, %r)\nfrom alproj_amd import launch\nc = launch.Control.from_env()\n"
                    "v = c.gather(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))\nc.close()\n"
                    "open(%r, 'w').write(repr(v)) if c.rank == 0 else None\n" % (ROOT, str(tmp_path / "seen")))
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    log = open(tmp_path / "log", "w+")
    assert launch.spawn([sys.executable, str(show)], 2, timeout_s=60, env=env, log=log) == 0
    assert (tmp_path / "seen").read_text() == "['0', '0']"
    log.seek(0)
    assert "HSA_ENABLE_IPC_MODE_LEGACY is unset" in log.read()
    log = open(tmp_path / "log2", "w+")
    assert launch.spawn([sys.executable, str(show)], 2, timeout_s=60, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="1"), log=log) == 0
    assert (tmp_path / "seen").read_text() == "['1', '1']"
    log.seek(0)
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in log.read()


def test_hub_collectives_in_process():
    n = 4
    hub = launch.Hub(n).start()
    res = [None] * n

    def rank(r):
        conn = launch.Control._connect(lambda: (hub.address, hub.authkey), r, 30.0)
        c = launch.Control(r, n, r, conn)
        c.barrier()
        got = (c.max(float(r)), c.bcast_bytes(bytes([r]) * 4), c.gather({"r": r}), c.bcast(("x", r)))
        c.close()
        res[r] = got

    th = [threading.Thread(target=rank, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    hub.join(5)
    assert hub.error is None and hub.collectives == 6
    for got in res:
        assert got == (3.0, b"\0\0\0\0", [{"r": i} for i in range(n)], ["x", 0])     # JSON on the wire: a tuple arrives as a list


def test_a_vanished_rank_releases_the_others():
    hub = launch.Hub(2).start()
    err = []

    def survivor():
        c = launch.Control(0, 2, 0, launch.Control._connect(lambda: (hub.address, hub.authkey), 0, 30.0))
        try:
            c.barrier()
        except launch.LaunchError as e:
            err.append(str(e))

    t = threading.Thread(target=survivor)
    t.start()
    gone = launch.Control._connect(lambda: (hub.address, hub.authkey), 1, 30.0)
    gone.close()                                               # dies before its first collective
    t.join(30)
    hub.join(5)
    assert err and "barrier failed" in err[0] and hub.error


def test_single_rank_control_is_the_identity():
    c = launch.Control()
    assert (c.rank, c.world) == (0, 1)
    c.barrier()
    assert c.max(2.5) == 2.5 and c.bcast_bytes(b"ab") == b"ab" and c.gather(7) == [7]
    c.close()


def test_host_digest_sees_every_word_and_ignores_the_thread_count():
    from alproj_amd import _lib
    rng = np.random.default_rng(0)
    a = rng.random(3_000_001)                                  # 24 MB: three slices, a ragged tail
    d = _lib.host_hash64(a)
    assert d == _lib.host_hash64(a, 1) == _lib.host_hash64(a, 3) == _lib.host_hash64(a.copy())
    for i in (0, 1, 1_048_575, 1_048_576, 2_345_678, len(a) - 1):
        b = a.copy()
        b[i] = np.nextafter(b[i], 2.0)                         # one bit of one word
        assert _lib.host_hash64(b) != d
    u = rng.integers(0, 256, 1001, dtype=np.uint8)
    du = _lib.host_hash64(u)
    for i in (0, 500, 999, 1000):                              # the tail bytes beyond the last whole word too
        v = u.copy()
        v[i] ^= 1
        assert _lib.host_hash64(v) != du
    assert _lib.host_hash64(u[:0]) != _lib.host_hash64(np.zeros(1, np.uint8))
    assert _lib.host_hash64(np.zeros(8, np.uint8)) != _lib.host_hash64(np.zeros(9, np.uint8))


def test_ranks_that_disagree_on_the_collective_are_told_so():
    hub = launch.Hub(2).start()
    errs = [None, None]

    def rank(r, op):
        c = launch.Control(r, 2, r, launch.Control._connect(lambda: (hub.address, hub.authkey), r, 30.0))
        try:
            c.barrier() if op == "barrier" else c.max(1.0)
        except launch.LaunchError as e:
            errs[r] = str(e)

    th = [threading.Thread(target=rank, args=(0, "barrier")), threading.Thread(target=rank, args=(1, "max"))]
    [t.start() for t in th]
    [t.join(30) for t in th]
    hub.join(5)
    assert all(e and "disagree" in e for e in errs) and "disagree" in hub.error


def test_spawn_returns_the_first_failing_code_and_zero_otherwise(tmp_path):
    """launch.spawn itself (what bench.py's launcher calls) on two tiny scripts"""
    ok = tmp_path / "ok.py"
    ok.write_text("import os, sys\nsys.path.insert(0, %r)\nfrom alproj_amd import launch\nc = launch.Control.from_env()\n"
                  "v = c.gather(c.rank)\nc.close()\nassert v == list(range(c.world))\n" % ROOT)
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nsys.exit(5) if os.environ['RANK'] == '2' else time.sleep(60)\n")
    assert launch.spawn([sys.executable, str(ok)], 4, timeout_s=60) == 0
    t0 = time.time()
    assert launch.spawn([sys.executable, str(bad)], 3, timeout_s=60, log=open(os.devnull, "w")) == 5
    assert time.time() - t0 < 30                                  # the sleeping ranks were ended, not waited for


def test_the_commands_of_the_first_multi_gpu_checklist_still_parse(monkeypatch):
    """README "First multi-GPU run": the five commands an operator is told to type must be ones bench.py accepts, the first
    one must do what the text says it does on a box without a GPU, and the table must be what bench.expectation_from_one_gpu
    derives from the committed 1-GPU line it names"""
    import re
    import bench
    text = open(os.path.join(ROOT, "README.md")).read()
    sec = text[text.index("## First multi-GPU run"):]
    block = sec[sec.index("```") + 3:]
    block = block[:block.index("```")]
    cmds = [re.sub(r"^\s*\d+\s+", "", line).split() for line in block.strip().splitlines()]
    assert len(cmds) == 5 and all(c[:2] == ["python", "bench.py"] for c in cmds)
    gpus = []
    for c in cmds:
        monkeypatch.setattr(sys, "argv", c[1:])
        args = bench.parse()                                   # argparse exits (SystemExit) on a flag that is gone
        gpus.append(args.gpus)
    assert gpus == [2, 2, 2, 2, 8]
    monkeypatch.setattr(sys, "argv", cmds[0][1:])
    assert bench.parse().launch_selftest
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + cmds[0][2:], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["selftest"] is True and line["n_gpus"] == 2 and len({x["uid_sha256"] for x in line["ranks"]}) == 1
    # the table's rows
    src = re.search(r"`(profiles/r0\d_bench_default_run.json)`", sec).group(1)
    for n, gpts, gens in re.findall(r"^\| (\d) \| (\d+)[^|]*\| ([\d.]+)[^|]*\|", sec, flags=re.M):
        if n == "1":
            continue
        one = json.load(open(os.path.join(ROOT, src)))
        k1, c1 = one["roofline"]["kernel_ms"], one["roofline"]["cma_kernel_ms"]
        host = one["cma"]["ms_per_iter"] - c1
        assert float(gpts) == pytest.approx(one["config"]["vertices"] / (k1 / int(n) / 1e3) / 1e9, rel=2e-3)
        assert float(gens) == pytest.approx(1e3 / (c1 / int(n) + 0.0045 + host), rel=6e-3)
