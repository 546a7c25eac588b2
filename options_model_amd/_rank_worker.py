"""One rank of a launcher.RankPool: `python -m options_model_amd._rank_worker DEVICE` with the rank environment
(RANK / WORLD_SIZE / MASTER_PORT / OMC_RDZV_NONCE) set by the parent.  Reads one JSON request per line from stdin,
runs it -- the same request arrives on every rank, the entry points are collective -- and answers with one JSON line
on stdout.  Everything else this process prints goes to stderr, so stdout carries the protocol only.

Requests: {"id": n, "fn": name, "kwargs": {...}} with fn one of
  __hello__                     bring the per-rank pricer (context + RCCL communicator) up; answers the transport
  price_american_option         api.price_american_option(**kwargs) inside the job (n_gpus = WORLD_SIZE)
  price_american_option_nn      the NN regressor sharded over the ranks (nn_dist.price_american_option_nn_sharded)
  __exit__                      leave (end of stdin does the same)
"""
from __future__ import annotations

import dataclasses
import json
import os
import sys


def _result_dict(res):
    d = dataclasses.asdict(res)
    plain = (int, float, str, bool, type(None))
    d["info"] = {k: v for k, v in d.get("info", {}).items()
                 if isinstance(v, plain) or (isinstance(v, list) and all(isinstance(x, plain) for x in v))}
    return d


def main(argv):
    device = int(argv[1]) if len(argv) > 1 else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ["WORLD_SIZE"])
    proto = os.fdopen(os.dup(sys.stdout.fileno()), "w", buffering=1)  # the protocol keeps the real stdout ...
    os.dup2(sys.stderr.fileno(), sys.stdout.fileno())                 # ... and stray prints land on stderr
    sys.stdout = sys.stderr

    def answer(msg):
        proto.write(json.dumps(msg) + "\n")
        proto.flush()

    from . import api
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        fn, kw, rid = req.get("fn"), req.get("kwargs", {}), req.get("id")
        if fn == "__exit__":
            break
        try:
            if fn == "__hello__":
                sp = api._job_pricer(world, device)
                out = dict(rank=sp.rank, world=sp.world, transport=sp.transport, pid=os.getpid())
            elif fn == "price_american_option":
                out = _result_dict(api.price_american_option(**dict(kw, n_gpus=world, device=device)))
            elif fn == "price_american_option_nn":
                from . import nn_dist
                out = _result_dict(nn_dist.price_american_option_nn_sharded(api._job_pricer(world, device), **kw))
            else:
                raise ValueError(f"unknown request {fn!r}")
            answer(dict(id=rid, ok=True, result=out))
        except Exception as e:  # reported to the parent, which decides (ValueError on all ranks: the pool lives on)
            answer(dict(id=rid, ok=False, type=type(e).__name__, error=str(e)))
            if not isinstance(e, ValueError):
                break  # this rank may be out of step with its peers: leave, the parent closes the pool
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
