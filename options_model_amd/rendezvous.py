"""Hand the RCCL unique id from rank 0 to the other ranks of ONE node without torch.distributed.

A communicator needs every rank to call omc_comm_init with the same 128-byte id that rank 0 drew
(ncclGetUniqueId).  All ranks of a job run on one node (one process per GPU: the job shape of
bench.py and of the driver's `python -m torch.distributed.run --nnodes=1 ...`), so the id travels
through a file in a directory all of them can see.  The file name carries MASTER_PORT and the id of
the ranks' common parent process (torchrun's agent, or bench.py's own launcher), so concurrent jobs
and earlier runs cannot collide; rank 0 publishes with an atomic rename and removes the file once its
own omc_comm_init has returned (the collective returns only after every rank has joined, i.e. read it).

The reference has no counterpart (no distributed code at all: SURVEY.md section 5.8).
"""
from __future__ import annotations

import os
import tempfile
import time


def _dirs() -> list:
    """Where the file may live, in order of preference; rank 0 takes the first one it can write to and
    the other ranks look in all of them."""
    out = []
    for d in (os.environ.get("OMC_RDZV_DIR"), "/dev/shm", tempfile.gettempdir()):
        if d and os.path.isdir(d) and d not in out:
            out.append(d)
    return out


def _name(tag: str | None = None) -> str:
    if tag is None:
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    return f"omc_rccl_uid_{tag}"


def publish(payload: bytes, tag: str | None = None) -> str:
    """Rank 0: make `payload` visible to the other ranks.  Returns the path (pass it to retire())."""
    last = None
    for d in _dirs():
        path = os.path.join(d, _name(tag))
        tmp = f"{path}.{os.getpid()}.tmp"
        try:
            with open(tmp, "wb") as f:
                f.write(payload)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, path)  # atomic: a reader sees the whole payload or no file
            return path
        except OSError as e:  # read-only or full: try the next directory
            last = e
    raise OSError(f"no writable rendezvous directory among {_dirs()}: {last}")


def fetch(nbytes: int, tag: str | None = None, timeout_s: float = 120.0) -> bytes:
    """Ranks > 0: wait for rank 0's payload.  Raises TimeoutError -- never hangs the job."""
    paths = [os.path.join(d, _name(tag)) for d in _dirs()]
    t0 = time.monotonic()
    while True:
        for path in paths:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == nbytes:
                    return data
            except OSError:
                pass
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
