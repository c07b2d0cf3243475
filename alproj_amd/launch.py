"""Torch-free launcher and control plane for one-process-per-GPU jobs on one node.

The data path of a multi-GPU job has exactly one collective -- the RCCL all-reduce of the P + 1 partial
sums inside libalproj_hip.so (the loop it shards is reference optimize.py:418-424).
Everything else a job needs between its ranks is control traffic: the 128-byte RCCL unique id from rank 0
to the others, barriers around timed regions, a max over the ranks' wall times, a few gathered records.
This module carries that over a localhost socket (``multiprocessing.connection``: stdlib; HMAC challenge on a
16-byte random key; messages are JSON bytes -- nothing received is ever unpickled),
so that neither the product nor ``bench.py`` needs ``torch.distributed``:

* ``spawn(argv, n)``   -- the PARENT: starts ``n`` fresh children of ``argv`` with ``RANK`` / ``LOCAL_RANK`` /
  ``WORLD_SIZE`` set, serves the hub, ends the others when one child fails, enforces one wall-clock limit.
  The parent never touches the GPU and never ``exec``s: children are ``subprocess.Popen``-ed before any HIP call.
* ``Control.from_env()`` -- a RANK: connects to the parent's hub (``ALPROJ_HUB``), or, when another launcher
  (``torch.distributed.run``) set ``WORLD_SIZE``, to a hub that rank 0 hosts and announces -- address AND a random
  key -- through a 0600 file in a directory only this user can enter (``$XDG_RUNTIME_DIR`` or a 0700 directory under
  the temporary directory, owner and mode checked), named after ``MASTER_ADDR`` / ``MASTER_PORT`` / the run id.
* ``gpu_nodes()`` / ``preflight(n)`` -- how many GPUs the node has, from sysfs (no HIP call), so that ``spawn`` is not
  asked for more ranks than there are devices.

A collective is one message per rank to the hub and one reply per rank: barrier, max, bcast (rank 0's
payload), gather (every rank's payload, in rank order).  A rank that dies closes its socket; the hub then
answers every other rank with an error, so nobody waits for a barrier that cannot complete.
"""
import glob
import hashlib
import json
import os
import signal
import stat
import subprocess
import sys
import tempfile
import threading
import time
from multiprocessing import AuthenticationError
from multiprocessing.connection import Client, Listener, answer_challenge, deliver_challenge

__all__ = ["Hub", "Control", "spawn", "LaunchError", "gpu_nodes", "preflight"]

HUB_ENV, KEY_ENV = "ALPROJ_HUB", "ALPROJ_HUB_KEY"
COLLECTIVE_TIMEOUT_S = 3600.0
HANDSHAKE_TIMEOUT_S = 10.0          # a peer that connects to the hub has this long to answer the key challenge


class _HandshakeTimeout(Exception):
    """a peer connected to the hub and did not answer the key challenge in time"""


class LaunchError(RuntimeError):
    pass


# ---------------------------------------------------------------------------------------------- wire format
# One message = one JSON document sent with send_bytes(): [op, payload] to the hub, [status, payload] back.  Payloads are
# None, numbers, strings, bytes (the RCCL id; carried as {"__bytes__": hex}), and lists / dicts of those.  Connection.send()
# would pickle -- and recv() unpickle whatever a peer that knows the key sends; this format cannot carry code.
def _enc(o):
    if isinstance(o, (bytes, bytearray)):
        return {"__bytes__": bytes(o).hex()}
    if isinstance(o, dict):
        return {str(k): _enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_enc(v) for v in o]
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    if hasattr(o, "item") and getattr(o, "shape", None) == ():      # a numpy scalar
        return o.item()
    raise TypeError(f"control-plane payloads are JSON values and bytes, not {type(o).__name__}")


def _dec(o):
    if isinstance(o, dict):
        if set(o) == {"__bytes__"}:
            return bytes.fromhex(o["__bytes__"])
        return {k: _dec(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_dec(v) for v in o]
    return o


def _send(conn, op, payload=None):
    conn.send_bytes(json.dumps([op, _enc(payload)]).encode())


def _recv(conn):
    """-> (op, payload); a malformed message is a LaunchError, never an object"""
    try:
        op, payload = json.loads(conn.recv_bytes(1 << 26).decode())
    except (ValueError, UnicodeDecodeError) as e:
        raise LaunchError(f"malformed control message ({e})") from e
    if not isinstance(op, str):
        raise LaunchError("malformed control message (no op)")
    return op, _dec(payload)


class Hub:
    """The meeting point of ``world`` ranks: accepts one connection per rank, then serves collectives until
    every rank has said goodbye (or one has gone away)."""

    def __init__(self, world, authkey=None, host="127.0.0.1"):
        self.world = int(world)
        self.authkey = authkey or os.urandom(16)
        # every rank knocks at once, before the serving thread exists (spawn starts the children first): room for all of them
        self.listener = Listener((host, 0), backlog=max(16, 2 * self.world), authkey=self.authkey)
        self.address = self.listener.address          # (host, port) actually bound
        self.error = None
        self.collectives = 0
        self._thread = None

    def start(self):
        self._thread = threading.Thread(target=self._serve, name="alproj-hub", daemon=True)
        self._thread.start()
        return self

    def join(self, timeout=None):
        if self._thread:
            self._thread.join(timeout)

    def close(self):
        try:
            self.listener.close()
        except OSError:
            pass

    def _accept_authenticated(self):
        """Listener.accept() with a deadline on the key handshake.  Listener's own runs the HMAC challenge on the accepted,
        BLOCKING connection: a local process that connects and then says nothing would park the hub in it for ever (the
        listening socket's timeout does not reach that far).  Here the challenge runs on a helper thread; a peer that has
        not answered within HANDSHAKE_TIMEOUT_S is hung up on."""
        try:
            raw = self.listener._listener.accept()            # the connection, no handshake yet
        except AttributeError:                                 # another Python's Listener internals: its own accept, as before
            return self.listener.accept()
        failure = []

        def handshake():
            try:
                deliver_challenge(raw, self.authkey)
                answer_challenge(raw, self.authkey)
            except BaseException as e:                         # noqa: BLE001 -- handed to the accepting thread below
                failure.append(e)

        t = threading.Thread(target=handshake, name="alproj-hub-handshake", daemon=True)
        t.start()
        t.join(HANDSHAKE_TIMEOUT_S)
        if t.is_alive():
            raw.close()                                        # the helper's recv fails and it ends
            t.join(1.0)
            raise _HandshakeTimeout()
        if failure:
            raw.close()
            if isinstance(failure[0], (AuthenticationError, EOFError, OSError)):
                raise AuthenticationError(str(failure[0]))
            raise failure[0]
        return raw

    def _serve(self):
        conns = [None] * self.world
        try:
            # the listening socket gets a timeout so that a rank which never starts does not park the hub for ever
            try:
                self.listener._listener._socket.settimeout(COLLECTIVE_TIMEOUT_S)
            except AttributeError:      # another Python's Listener internals: the launcher's wall-clock limit still holds
                pass
            joined = 0
            while joined < self.world:
                try:
                    c = self._accept_authenticated()
                except TimeoutError:
                    raise LaunchError(f"only {joined} of {self.world} ranks reached the hub within {COLLECTIVE_TIMEOUT_S:.0f} s") from None
                except (AuthenticationError, EOFError, ConnectionError, _HandshakeTimeout):
                    continue            # a stranger, a silent one, or a rank of another job with that job's key: not ours, keep listening
                joined += 1
                op, rank = _recv(c)
                if op != "hello" or not isinstance(rank, int) or not (0 <= rank < self.world) or conns[rank] is not None:
                    raise LaunchError(f"unexpected greeting {op!r} from rank {rank!r}")
                conns[rank] = c
            for c in conns:
                _send(c, "ok", self.world)
            alive = self.world
            while alive:
                msgs = []
                for r, c in enumerate(conns):
                    if not c.poll(COLLECTIVE_TIMEOUT_S):
                        raise LaunchError(f"rank {r} did not reach collective {self.collectives} within {COLLECTIVE_TIMEOUT_S:.0f} s")
                    msgs.append(_recv(c))              # EOFError when the rank has gone away
                ops = {m[0] for m in msgs}
                if len(ops) != 1:
                    raise LaunchError(f"ranks disagree on collective {self.collectives}: {sorted(ops)}")
                op = ops.pop()
                payloads = [m[1] for m in msgs]
                if op == "barrier":
                    out = [None] * self.world
                elif op == "max":
                    out = [max(float(x) for x in payloads)] * self.world
                elif op == "bcast":
                    out = [payloads[0]] * self.world
                elif op == "gather":
                    out = [payloads] * self.world
                elif op == "bye":
                    out = [None] * self.world
                    alive = 0
                else:
                    raise LaunchError(f"unknown collective {op!r}")
                for c, o in zip(conns, out):
                    _send(c, "ok", o)
                self.collectives += 1
        except (EOFError, OSError, LaunchError, TypeError, ValueError) as e:
            self.error = f"{type(e).__name__}: {e}" if str(e) else f"{type(e).__name__}: a rank closed its connection"
            for c in conns:
                if c is not None:
                    try:
                        _send(c, "error", self.error)
                    except (OSError, ValueError):
                        pass
        finally:
            for c in conns:
                if c is not None:
                    try:
                        c.close()
                    except OSError:
                        pass
            self.close()


def _private_dir():
    """A directory only this user can enter: $XDG_RUNTIME_DIR when it is one, else <tmp>/alproj_<uid> made 0700.  An
    existing directory is accepted only if it is a real directory (not a symlink), owned by this user, with no access
    for anybody else -- in a sticky /tmp somebody else may have made a directory of that name first."""
    uid = os.getuid()
    cands = [os.environ.get("XDG_RUNTIME_DIR"), os.path.join(tempfile.gettempdir(), f"alproj_{uid}")]
    problems = []
    for d in cands:
        if not d:
            continue
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
        except OSError as e:
            problems.append(f"{d}: {e}")
            continue
        try:
            st = os.lstat(d)
        except OSError as e:
            problems.append(f"{d}: {e}")
            continue
        if stat.S_ISDIR(st.st_mode) and st.st_uid == uid and not (st.st_mode & 0o077):
            return d
        problems.append(f"{d}: not a private directory of uid {uid} (mode {stat.S_IMODE(st.st_mode):o}, owner {st.st_uid})")
    raise LaunchError("no private directory for the hub announcement: " + "; ".join(problems))


def _rendezvous_file():
    """Where rank 0 announces its hub when a foreign launcher started the ranks: one name per job on this node, inside
    the private directory.  The name is public knowledge (MASTER_ADDR / MASTER_PORT / run id); the KEY is not derived
    from it -- rank 0 draws it and it travels inside the file."""
    tag = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK"))
    return os.path.join(_private_dir(), "hub_" + hashlib.sha256(tag.encode()).hexdigest()[:20])


def _announce(path, address, key):
    """rank 0: publish address and key.  A stale file of an earlier job is removed first (it lives in a directory nobody
    else can write to); the new one is created exclusively, never through a symlink, readable by this user only."""
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    with os.fdopen(fd, "w") as f:
        json.dump({"host": address[0], "port": address[1], "key": key.hex()}, f)


def _read_announcement(path):
    """a rank: ((host, port), key) from rank 0's file -- only if the file is a regular file of this user with mode 0600"""
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))        # FileNotFoundError until rank 0 has written it
    with os.fdopen(fd) as f:
        st = os.fstat(f.fileno())
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise LaunchError(f"{path}: not a private file of uid {os.getuid()}")
        d = json.load(f)
    return (str(d["host"]), int(d["port"])), bytes.fromhex(d["key"])


class Control:
    """A rank's end of the control plane.  ``world == 1`` needs no hub: every collective is the identity."""

    def __init__(self, rank=0, world=1, local_rank=0, conn=None, hub=None, announce=None):
        self.rank, self.world, self.local_rank = int(rank), int(world), int(local_rank)
        self._conn, self._hub, self._announce = conn, hub, announce

    @classmethod
    def from_env(cls, connect_timeout_s=300.0):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world <= 1:
            return cls(0, 1, local_rank)
        hub = announce = None
        if HUB_ENV in os.environ:                       # started by spawn(): the parent serves the hub
            host, port = os.environ[HUB_ENV].rsplit(":", 1)
            address, key = (host, int(port)), bytes.fromhex(os.environ[KEY_ENV])
            conn = cls._connect(lambda: (address, key), rank, connect_timeout_s)
        else:                                           # started by torch.distributed.run or the like
            path = _rendezvous_file()
            if rank == 0:
                hub = Hub(world).start()                # a fresh 16-byte random key
                _announce(path, hub.address, hub.authkey)
                announce = path
            conn = cls._connect(lambda: _read_announcement(path), rank, connect_timeout_s)
        return cls(rank, world, local_rank, conn, hub, announce)

    @staticmethod
    def _connect(where, rank, timeout_s):
        """A stale announcement (an earlier job with the same MASTER_PORT) points at a closed port or carries a key the
        listener there does not share: both are retried until rank 0's new announcement is there."""
        t0 = time.time()
        last = None
        while time.time() - t0 < timeout_s:
            try:
                address, key = where()
                conn = Client(address, authkey=key)
                _send(conn, "hello", rank)
                status, _ = _recv(conn)
                if status == "ok":
                    return conn
                last = LaunchError("hub refused the greeting")
            except (OSError, EOFError, ValueError, Exception) as e:      # noqa: B014 -- AuthenticationError is a ProcessError
                last = e
            time.sleep(0.05)
        raise LaunchError(f"rank {rank}: no hub after {timeout_s:.0f} s ({last})")

    def _collective(self, op, payload=None):
        if self.world <= 1:
            return {"barrier": None, "max": payload, "bcast": payload, "gather": [payload], "bye": None}[op]
        try:
            _send(self._conn, op, payload)
            status, out = _recv(self._conn)
        except (EOFError, OSError) as e:
            raise LaunchError(f"rank {self.rank}: the hub went away during {op} ({type(e).__name__})") from e
        if status != "ok":
            raise LaunchError(f"rank {self.rank}: {op} failed: {out}")
        return out

    def barrier(self):
        self._collective("barrier")

    def max(self, x):
        return self._collective("max", float(x))

    def bcast_bytes(self, b):
        return self._collective("bcast", bytes(b))

    def bcast(self, obj):
        return self._collective("bcast", obj)

    def gather(self, obj):
        return self._collective("gather", obj)

    def close(self):
        if self.world > 1 and self._conn is not None:
            try:
                self._collective("bye")
            finally:
                self._conn.close()
                self._conn = None
        if self._hub is not None:
            self._hub.join(5.0)
            self._hub = None
        if self._announce:
            try:
                os.unlink(self._announce)
            except OSError:
                pass
            self._announce = None


try:        # resolved HERE, in the parent: between fork and exec a child must not import or dlopen anything
    import ctypes as _ctypes
    _prctl = _ctypes.CDLL("libc.so.6", use_errno=True).prctl
except Exception:       # not Linux / no libc by that name: ranks then notice a dead launcher at their next collective
    _prctl = None
_SIGTERM = int(signal.SIGTERM)


def _die_with_parent():
    """preexec of a child: SIGTERM when the parent goes away (Linux PR_SET_PDEATHSIG = 1), so that no rank outlives a
    killed launcher and keeps a GPU busy.  One call of an already loaded C function, nothing else."""
    if _prctl is not None:
        _prctl(1, _SIGTERM)


def _stop(procs, grace_s=5.0):
    """End exactly the processes this launcher started (by PID, never by pattern)."""
    for p in procs:
        if p.poll() is None:
            try:
                p.terminate()
            except OSError:
                pass
    t0 = time.time()
    while time.time() - t0 < grace_s and any(p.poll() is None for p in procs):
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
KFD_ENV = "ALPROJ_KFD_TOPOLOGY"          # another root for the same tree (tests)


def gpu_nodes(root=None):
    """GPUs of this node as the kernel driver lists them -- KFD topology nodes with ``simd_count > 0`` (CPUs are nodes
    with none) -- read from sysfs: no HIP call, so the launcher stays a process that never initialised the GPU.
    Returns (count, what was read); count is None when the tree is absent (no amdgpu driver, or not Linux)."""
    root = root or os.environ.get(KFD_ENV) or KFD_NODES
    files = sorted(glob.glob(os.path.join(root, "*", "properties")))
    if not files:
        return None, root
    n = 0
    for f in files:
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            continue
    return n, root


def _visible_limit(env):
    """devices left by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (None: no restriction)"""
    lim = None
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(k)
        if v is not None:
            c = len([x for x in v.split(",") if x.strip() != ""])
            lim = c if lim is None else min(lim, c)
    return lim


def preflight(n, env=None, log=sys.stderr, root=None):
    """True when ``n`` ranks can each have a GPU of their own.  Says in ONE line what is wrong otherwise."""
    env = os.environ if env is None else env
    have, where = gpu_nodes(root)
    if have is None:
        print(f"launch: no GPU on this node: {where} has no topology nodes (is the amdgpu driver loaded?); --gpus {n} needs {n}",
              file=log, flush=True)
        return False
    lim = _visible_limit(env)
    usable = have if lim is None else min(have, lim)
    if usable < n:
        why = f"{have} GPU node(s) under {where}" + (f", {lim} left by *_VISIBLE_DEVICES" if lim is not None and lim < have else "")
        print(f"launch: --gpus {n} but this node offers {usable}: {why}; nothing was started", file=log, flush=True)
        return False
    return True


def spawn(argv, n, timeout_s=3600.0, env=None, log=sys.stderr):
    """Run ``n`` ranks of ``argv`` and return the job's exit code: 0 when every rank returned 0; the first failing
    rank's code (the others are ended) otherwise; 124 when ``timeout_s`` of wall clock ran out.

    Rank r gets ``RANK=r LOCAL_RANK=r WORLD_SIZE=n`` and the hub's address; rank 0 inherits stdout (the ONE JSON
    line), every rank inherits stderr."""
    n = int(n)
    if n < 1:
        raise ValueError("need at least one rank")
    hub = Hub(n)
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), HUB_ENV: f"{hub.address[0]}:{hub.address[1]}", KEY_ENV: hub.authkey.hex()})
    # HSA_ENABLE_IPC_MODE_LEGACY is the operator's to set: a value in the environment is passed on untouched.  Only when it
    # is absent do the ranks get 0 -- and a line saying so: the operating note of the pool this was built on states that its
    # host driver supports dmabuf IPC only and that RCCL fails with "hipIpcGetMemHandle: invalid argument" without it
    # (the note exports the variable itself on every box; no multi-GPU run of this code exists yet to confirm either way --
    # alp_comm_init's failure message prints the value in force).
    if n > 1 and "HSA_ENABLE_IPC_MODE_LEGACY" not in base:
        base["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        print("launch: HSA_ENABLE_IPC_MODE_LEGACY is unset; the ranks get 0 (dmabuf IPC; set it yourself to override)", file=log, flush=True)
    procs = []
    try:
        for r in range(n):       # children first, hub thread afterwards: no fork out of a threaded parent
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL,
                                          preexec_fn=_die_with_parent))
        hub.start()
        t0 = time.time()
        rc = None
        while rc is None:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"launch: rank {r} (pid {procs[r].pid}) exited with {c}; ending the other ranks", file=log, flush=True)
                rc = c if c > 0 else 128 - c            # a signal's negative code as the shell would show it
            elif all(c == 0 for c in codes):
                rc = 0
            elif time.time() - t0 > timeout_s:
                print(f"launch: {timeout_s:.0f} s of wall clock are over; ending all ranks", file=log, flush=True)
                rc = 124
            else:
                time.sleep(0.05)
        return rc
    finally:
        _stop(procs)
        hub.close()
        hub.join(2.0)
