"""Hand the RCCL unique id from rank 0 to the other ranks of ONE node without torch.distributed, and let
the ranks agree on a yes/no question (did everybody's communicator come up?) the same way.

A communicator needs every rank to call omc_comm_init with the same 128-byte id that rank 0 drew
(ncclGetUniqueId).  All ranks of a job run on one node (one process per GPU: the job shape of
bench.py and of the driver's `python -m torch.distributed.run --nnodes=1 ...`), so the id travels
through a file in a directory all of them can see.

Names.  The file name carries MASTER_PORT, the id of the ranks' common parent process (torchrun's
agent, or bench.py's own launcher) and -- when the launcher exported one -- a per-launch random nonce
(OMC_RDZV_NONCE), so concurrent jobs and earlier runs cannot collide.

Files.  Created with O_CREAT|O_EXCL|O_NOFOLLOW and mode 0600 (never through an existing name or a
symlink, not readable by other users), written completely under a temporary name and renamed into place
(a reader sees the whole payload or no file).  Every payload is framed: 8 bytes magic, 8 bytes creation
time, the tag's hash, then the body; fetch() accepts only a frame whose tag hash matches and whose
creation time is not older than the reader's own process start (minus a slack), so a file left behind by
a killed run under the same name is never taken for the current one.  Rank 0 removes whatever sits under
its name before it publishes and retires the file once its own omc_comm_init has returned.

The reference has no counterpart (no distributed code at all: SURVEY.md section 5.8).
"""
from __future__ import annotations

import hashlib
import os
import struct
import tempfile
import time

MAGIC = b"OMCRDZV2"
_HEADER = len(MAGIC) + 8 + 16
STALE_SLACK_S = 30.0  # ranks of one launch start within seconds of each other


def _process_start() -> float:
    """Wall-clock start time of THIS process: psutil, else /proc (start time in clock ticks since boot + the boot
    time), else the import time of this module -- which can be long after the start (a rank that imports this module
    more than STALE_SLACK_S after rank 0 published would reject rank 0's frame as stale), hence only the last resort."""
    try:
        import psutil
        return float(psutil.Process().create_time())
    except Exception:
        pass
    try:
        with open("/proc/self/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])  # field 22 (starttime); the comm field may hold spaces
        with open("/proc/stat") as f:
            btime = next(float(ln.split()[1]) for ln in f if ln.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return _IMPORT_TIME


def _not_before() -> float:
    """Frames older than this are leftovers of an earlier launch.  A launch that carries a nonce (launcher.py,
    bench.py, torchrun's run id) has file names no other launch can produce: nothing to reject there."""
    nonce = os.environ.get("OMC_RDZV_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID")
    if nonce and nonce != "none":
        return 0.0
    return _process_start() - STALE_SLACK_S


_IMPORT_TIME = time.time()


def _dirs() -> list:
    """Where the file may live, in order of preference; rank 0 takes the first one it can write to and
    the other ranks look in all of them."""
    out = []
    for d in (os.environ.get("OMC_RDZV_DIR"), "/dev/shm", tempfile.gettempdir()):
        if d and os.path.isdir(d) and d not in out:
            out.append(d)
    return out


def default_tag() -> str:
    tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    nonce = os.environ.get("OMC_RDZV_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID")
    if nonce and nonce != "none":
        tag += "_" + "".join(ch for ch in nonce if ch.isalnum())[:32]
    return tag


def _nonce() -> str:
    nonce = os.environ.get("OMC_RDZV_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID")
    return "".join(ch for ch in nonce if ch.isalnum())[:32] if nonce and nonce != "none" else ""


def _name(tag: str | None = None) -> str:
    """File name of a rendezvous: the default tag carries the launch nonce already; an EXPLICIT tag gets it appended, so
    that under a nonce no file name can be produced by another launch whatever tag the caller chose -- which is what
    lets _not_before() skip the staleness test there."""
    if tag is None:
        return f"omc_rccl_uid_{default_tag()}"
    n = _nonce()
    return f"omc_rccl_uid_{tag}" + (f"_{n}" if n and n not in tag else "")


def _tag_hash(tag: str | None) -> bytes:
    return hashlib.sha256(_name(tag).encode()).digest()[:16]


def _frame(payload: bytes, tag: str | None) -> bytes:
    return MAGIC + struct.pack("<d", time.time()) + _tag_hash(tag) + payload


def _unframe(data: bytes, nbytes: int, tag: str | None, not_before: float):
    """-> payload, or None when `data` is not a complete, current frame for this tag."""
    if len(data) != _HEADER + nbytes or not data.startswith(MAGIC):
        return None
    (created,) = struct.unpack("<d", data[len(MAGIC):len(MAGIC) + 8])
    if data[len(MAGIC) + 8:_HEADER] != _tag_hash(tag) or created < not_before:
        return None
    return data[_HEADER:]


def _write_private(path: str, data: bytes) -> None:
    """Create `path` afresh (O_EXCL, no symlink following, 0600) with `data`, atomically."""
    tmp = f"{path}.{os.getpid()}.tmp"
    flags = os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0)
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, flags, 0o600)
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(data)
            f.flush()
            os.fsync(f.fileno())
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise
    os.replace(tmp, path)  # a reader sees the whole frame or no file (rename never follows `path`)


def _read_regular(path: str):
    """Contents of `path` if it is a regular file that can be opened without following a symlink."""
    try:
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except OSError:
        return None
    try:
        with os.fdopen(fd, "rb") as f:
            return f.read()
    except OSError:
        return None


def publish(payload: bytes, tag: str | None = None) -> str:
    """Rank 0: make `payload` visible to the other ranks.  Returns the path (pass it to retire())."""
    last = None
    for d in _dirs():
        path = os.path.join(d, _name(tag))
        try:
            try:
                os.unlink(path)  # whatever an earlier run left under this name
            except FileNotFoundError:
                pass
            _write_private(path, _frame(payload, tag))
            return path
        except OSError as e:  # read-only or full: try the next directory
            last = e
    raise OSError(f"no writable rendezvous directory among {_dirs()}: {last}")


def fetch(nbytes: int, tag: str | None = None, timeout_s: float = 120.0) -> bytes:
    """Ranks > 0: wait for rank 0's payload.  Raises TimeoutError -- never hangs the job."""
    paths = [os.path.join(d, _name(tag)) for d in _dirs()]
    not_before = _not_before()
    t0 = time.monotonic()
    while True:
        for path in paths:
            data = _read_regular(path)
            if data is not None:
                payload = _unframe(data, nbytes, tag, not_before)
                if payload is not None:
                    return payload
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"rank 0 never published {paths[0]} within {timeout_s:.0f} s")
        time.sleep(0.005)


def retire(path: str) -> None:
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass


def exchange(rank: int, make_payload, nbytes: int, tag: str | None = None, timeout_s: float = 120.0):
    """-> (payload, path_or_None).  Rank 0 calls make_payload() and publishes; the rest fetch."""
    if rank == 0:
        payload = make_payload()
        assert len(payload) == nbytes
        return payload, publish(payload, tag)
    return fetch(nbytes, tag, timeout_s), None


# ------------------------------------------------------------------ a collective yes / no
def _collect(base: str, kind: str, world: int, phase: str, timeout_s: float) -> dict:
    got, missing = {}, set(range(world))
    not_before = _not_before()
    t0 = time.monotonic()
    while missing:
        for r in sorted(missing):
            t = f"{base}_{kind}{r}"
            for d in _dirs():
                data = _read_regular(os.path.join(d, _name(t)))
                body = _unframe(data, 1, t, not_before) if data is not None else None
                if body is not None:
                    got[r] = body
                    missing.discard(r)
                    break
        if missing and time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"rendezvous '{phase}': rank(s) {sorted(missing)} never reported within "
                               f"{timeout_s:.0f} s")
        if missing:
            time.sleep(0.005)
    return got


def agree(rank: int, world: int, ok: bool, phase: str, tag: str | None = None, timeout_s: float = 120.0) -> bool:
    """Every rank reports `ok` for `phase`; returns True iff ALL ranks reported True.  Raises TimeoutError naming
    the ranks that never reported -- a rank that died or hangs before this point therefore turns into an error
    on every other rank instead of a wait without end.  Each rank writes one small vote file and reads the
    others'.  A vote must outlive its writer's interest in it (a rank that votes and exits at once must still be
    heard), so it is removed only after a second round of files has told its writer that every rank has read the
    votes; that round is best effort (short wait), its files go when the process exits."""
    base = (default_tag() if tag is None else tag) + f"_{phase}"
    mine = publish(b"\x01" if ok else b"\x00", f"{base}_v{rank}")
    votes = _collect(base, "v", world, phase, timeout_s)
    verdict = all(v == b"\x01" for v in votes.values())
    _retire_at_exit([publish(b"\x01", f"{base}_a{rank}")])
    try:
        _collect(base, "a", world, phase, min(5.0, timeout_s))
        retire(mine)  # everybody has read every vote
    except TimeoutError:
        pass          # somebody is slow or gone: the few bytes stay rather than being missed
    return verdict


_exit_paths: list = []


def _retire_at_exit(paths) -> None:
    if not _exit_paths:
        import atexit
        atexit.register(lambda: [retire(p) for p in _exit_paths])
    _exit_paths.extend(paths)
