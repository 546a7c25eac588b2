"""Rank processes for `n_gpus = N` calls made from a PLAIN process (one process per GPU, started here).

The reference's callers are single-process programs -- the Streamlit scripts (options_model_2_ui.py:87-133) and the
spawn pool of options_model_3.py:1043-1056 -- so the drop-in cannot ask them to be started by a launcher.  Two forms:

  launch_ranks(n, deadline_s, argv)   one-shot: N children run `argv` with the rank environment, rank 0's stdout is
                                      relayed (bench.py --gpus N without a launcher);
  RankPool(n)                         persistent: N workers (`python -m options_model_amd._rank_worker`) bring their
                                      RCCL communicator up once and then serve calls; api.price_american_option and
                                      AdvancedOptionPricer use the process-wide pool of `pool(n)`.

Rules both follow (GPU boxes take a machine down for less):
  * the parent makes no HIP call here and NEVER replaces itself (`os.exec*`): ranks are CHILD processes started with
    subprocess (fresh interpreters: they initialise the GPU themselves, after their own start);
  * every wait has a deadline; on a dead rank or a passed deadline exactly the processes started here are terminated
    (by pid -- never by pattern) and the call fails with the ranks named;
  * a failing rank takes the job down: nothing is retried, nothing falls back to fewer GPUs.
"""
from __future__ import annotations

import atexit
import json
import os
import selectors
import socket
import subprocess
import sys
import threading
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, nonce: str, base=None, local_rank=None) -> dict:
    """Environment of rank `rank` of a `world`-rank job on this node (what torch.distributed.run would set, plus the
    nonce that keeps this launch's rendezvous files apart from any other launch's)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank if local_rank is None else local_rank), WORLD_SIZE=str(world),
               LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMC_RDZV_NONCE=nonce)
    # the host driver only supports dmabuf IPC: RCCL needs this (exported on the GPU boxes; set it for launches from
    # a bare environment too, never override what the caller chose)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _stderr_fd():
    """The parent's stderr for the children (None = inherit fd 2 when sys.stderr is not a real file, e.g. captured)."""
    try:
        return sys.stderr.fileno()
    except (AttributeError, OSError, ValueError):
        return None


def _stop(procs, ranks, grace_s: float = 10.0):
    for q in ranks:  # exactly the processes started by this module
        if procs[q].poll() is None:
            procs[q].terminate()
    t1 = time.monotonic()
    for q in ranks:
        try:
            procs[q].wait(max(0.1, grace_s - (time.monotonic() - t1)))
        except subprocess.TimeoutExpired:
            procs[q].kill()
            procs[q].wait()


def launch_ranks(n: int, deadline_s: float, argv, who: str = "options_model_amd.launcher",
                 timeout_flag: str = "the deadline") -> int:
    """Start N rank processes running `argv` and relay rank 0's stdout.  A child that fails takes the job down with a
    non-zero exit code (children are never re-exec'd or retried), and so does the overall deadline: the ranks still
    running then are terminated -- exactly the processes started here -- and named on stderr, so that a rank stuck
    inside a communicator call ends as an error, not as a job that never finishes."""
    import tempfile
    port = free_port()
    nonce = os.urandom(8).hex()
    procs = []
    out0 = tempfile.TemporaryFile()  # rank 0's stdout (a file, so that nobody blocks on a full pipe)
    for r in range(n):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, n, port, nonce),
                                      stdout=out0 if r == 0 else _stderr_fd()))
    rc = 0
    alive = set(range(n))
    t0 = time.monotonic()
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"{who}: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                _stop(procs, sorted(alive))
        if alive and rc == 0 and time.monotonic() - t0 > deadline_s:
            rc = 124
            print(f"{who}: rank(s) {sorted(alive)} did not finish within {timeout_flag} {deadline_s:.0f} s; "
                  f"terminating them", file=sys.stderr)
            _stop(procs, sorted(alive))
        if alive:
            time.sleep(0.05)
    out0.seek(0)
    out = out0.read().decode()
    if rc == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
    else:
        sys.stderr.write(out)
    return rc


# How long the other ranks may take to raise the same ValueError once the first rank has (see _roundtrip): at least
# VALUE_ERROR_GRACE_S, and VALUE_ERROR_GRACE_FRACTION of the call's own timeout -- a ValueError that every rank raises
# legitimately can arrive with seconds of skew on a first call (one rank still building or loading the library, creating
# its context, or generating a longer shard before a library check turns a negative return code into the exception), and
# five seconds flat would have declared those ranks out of step and cost the pool.
VALUE_ERROR_GRACE_S = 5.0
VALUE_ERROR_GRACE_FRACTION = 0.05  # 30 s at the default 600 s timeout


class RankError(RuntimeError):
    """A rank of the pool died, answered with an error, or did not answer within the deadline.  The pool is closed."""


class RankPool:
    """N worker processes, one per GPU, alive across calls.  `call(fn, kwargs)` sends the same request to every rank
    (the multi-GPU entry points are collective), waits for every rank's answer and returns rank 0's.

    devices: the HIP device of each rank (default: rank r -> device r).  env: extra environment for the workers
    (tests point OMC_RCCL_LIB at the shared-memory stand-in to run several ranks on one card).  worker_argv: the
    worker command (default: this package's _rank_worker; the CPU tests run a protocol-only worker)."""

    def __init__(self, n_gpus: int, devices=None, env=None, start_timeout_s: float = 300.0, worker_argv=None):
        n = int(n_gpus)
        if n < 2:
            raise ValueError("a rank pool needs at least two ranks")
        self.n = n
        self.devices = [int(d) for d in (devices if devices is not None else range(n))]
        if len(self.devices) != n:
            raise ValueError("one device per rank")
        self._lock = threading.Lock()
        self._seq = 0
        self._bufs = [b""] * n
        self.procs = []
        port, nonce = free_port(), os.urandom(8).hex()
        base = dict(os.environ)
        base.update(env or {})
        base["PYTHONPATH"] = _ROOT + os.pathsep + base.get("PYTHONPATH", "") if base.get("PYTHONPATH") else _ROOT
        try:
            for r in range(n):
                self.procs.append(subprocess.Popen(
                    list(worker_argv or [sys.executable, "-m", "options_model_amd._rank_worker"]) + [str(self.devices[r])],
                    env=rank_env(r, n, port, nonce, base), stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                    stderr=_stderr_fd(), cwd=_ROOT))
            for p in self.procs:
                os.set_blocking(p.stdout.fileno(), False)
            self._closed = False
            # first exchange: every worker has imported the package, created its context and communicator
            hello = self._roundtrip(dict(fn="__hello__", kwargs={}), start_timeout_s)
            bad = [(r, a) for r, a in enumerate(hello) if not a.get("ok")]
            if bad:
                raise RankError("the ranks did not come up: " +
                                "; ".join(f"rank {r}: {a.get('type')}: {a.get('error')}" for r, a in bad))
            self.transport = hello[0]["result"].get("transport")
        except BaseException:
            self._closed = False
            self.close(kill=True)
            raise

    # ---------------------------------------------------------------- protocol: one JSON object per line
    def _roundtrip(self, req: dict, timeout_s: float) -> list:
        self._seq += 1
        req = dict(req, id=self._seq)
        line = (json.dumps(req) + "\n").encode()
        for r, p in enumerate(self.procs):
            try:
                p.stdin.write(line)
                p.stdin.flush()
            except (BrokenPipeError, OSError) as e:
                raise RankError(f"rank {r} is gone (exit code {p.poll()}): {e}") from None
        answers = [None] * self.n
        sel = selectors.DefaultSelector()
        for r, p in enumerate(self.procs):
            sel.register(p.stdout, selectors.EVENT_READ, r)
        deadline = time.monotonic() + timeout_s
        refused_at = None  # when the first rank answered ValueError
        try:
            while any(a is None for a in answers):
                left = deadline - time.monotonic()
                if left <= 0:
                    missing = [r for r, a in enumerate(answers) if a is None]
                    raise RankError(f"rank(s) {missing} did not answer {req['fn']} within {timeout_s:.0f} s")
                if refused_at is not None and time.monotonic() - refused_at > max(VALUE_ERROR_GRACE_S,
                                                                                      VALUE_ERROR_GRACE_FRACTION * timeout_s):
                    # An argument error is raised before anything collective -- on EVERY rank, within moments.  A rank
                    # that has not refused by now took the call: the ValueError was rank-local (a library check that only
                    # fails for one shard), and that rank's peers sit in a collective it will never enter.
                    bad = [r for r, a in enumerate(answers) if a is not None and not a.get("ok")]
                    missing = [r for r, a in enumerate(answers) if a is None]
                    raise RankError(f"rank(s) {bad} refused {req['fn']} ({answers[bad[0]].get('error')}) while rank(s) "
                                    f"{missing} went ahead: the ranks are out of step")
                for key, _ in sel.select(min(left, 0.5)):
                    r = key.data
                    chunk = self.procs[r].stdout.read()
                    if chunk:
                        self._bufs[r] += chunk
                    elif chunk == b"":  # end of file: the worker is gone
                        raise RankError(f"rank {r} exited (code {self.procs[r].wait()}) during {req['fn']}")
                    while b"\n" in self._bufs[r]:
                        ln, self._bufs[r] = self._bufs[r].split(b"\n", 1)
                        if not ln.startswith(b"{"):
                            continue  # stray prints of a library: not part of the protocol
                        msg = json.loads(ln)
                        if msg.get("id") == req["id"]:
                            answers[r] = msg
                            if not msg.get("ok") and msg.get("type") == "ValueError" and refused_at is None:
                                refused_at = time.monotonic()
                            if not msg.get("ok") and msg.get("type") != "ValueError":
                                # this rank has left the collective call: its peers may be waiting for it inside a
                                # collective that has no deadline -- do not wait for their answers
                                raise RankError(f"rank {r}: {msg.get('type')}: {msg.get('error')}")
                for r, p in enumerate(self.procs):
                    if answers[r] is None and p.poll() is not None:
                        raise RankError(f"rank {r} exited (code {p.returncode}) during {req['fn']}")
        finally:
            sel.close()
        return answers

    def call(self, fn: str, kwargs: dict, timeout_s: float = 600.0):
        return self.call_all(fn, kwargs, timeout_s)[0]

    def call_all(self, fn: str, kwargs: dict, timeout_s: float = 600.0):
        """-> every rank's result (dicts, rank order); call() returns rank 0's.  An error raised on the ranks is re-raised here: ValueError as ValueError
        (every rank validates the same arguments), anything else as RankError -- after which the pool is closed,
        because ranks that failed at different points are no longer in step."""
        with self._lock:
            if self._closed:
                raise RankError("the rank pool is closed")
            try:
                answers = self._roundtrip(dict(fn=fn, kwargs=kwargs), timeout_s)
            except BaseException:
                self.close(kill=True)
                raise
            bad = [(r, a) for r, a in enumerate(answers) if not a.get("ok")]
            if bad:
                if len(bad) == self.n and all(a.get("type") == "ValueError" for _, a in bad):
                    raise ValueError(bad[0][1].get("error", "invalid argument"))  # nothing collective was entered
                self.close(kill=True)
                raise RankError("; ".join(f"rank {r}: {a.get('type')}: {a.get('error')}" for r, a in bad))
            return [a["result"] for a in answers]

    def close(self, kill: bool = False):
        if getattr(self, "_closed", True):
            return
        self._closed = True
        if not kill:
            for p in self.procs:  # an orderly exit: workers destroy their communicator collectively
                try:
                    p.stdin.write(b'{"fn": "__exit__", "id": 0, "kwargs": {}}\n')
                    p.stdin.flush()
                    p.stdin.close()
                except (BrokenPipeError, OSError, ValueError):
                    pass
            t0 = time.monotonic()
            for p in self.procs:
                try:
                    p.wait(max(0.1, 20.0 - (time.monotonic() - t0)))
                except subprocess.TimeoutExpired:
                    pass
        _stop(self.procs, range(len(self.procs)), grace_s=5.0)
        for p in self.procs:
            for f in (p.stdin, p.stdout):
                try:
                    if f:
                        f.close()
                except (OSError, ValueError):
                    pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_pools = {}
_pools_lock = threading.Lock()


def pool(n_gpus: int, devices=None, env=None) -> RankPool:
    """The process-wide pool for n_gpus ranks (started on first use, reused by later calls, closed at exit)."""
    key = (os.getpid(), int(n_gpus), tuple(devices) if devices is not None else None)
    with _pools_lock:
        p = _pools.get(key)
        if p is None or p._closed:
            p = RankPool(n_gpus, devices=devices, env=env)
            _pools[key] = p
        return p


def close_pools():
    with _pools_lock:
        for p in list(_pools.values()):
            p.close()
        _pools.clear()


atexit.register(close_pools)
