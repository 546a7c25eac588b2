"""Protocol-only stand-in for options_model_amd._rank_worker (CPU tests of launcher.RankPool): no GPU, no library."""
import json
import os
import sys
import time

device = sys.argv[1] if len(sys.argv) > 1 else "?"
rank = int(os.environ["RANK"])
for line in sys.stdin:
    req = json.loads(line)
    fn, kw, rid = req["fn"], req.get("kwargs", {}), req["id"]
    if fn == "__exit__":
        break
    if fn == "__hello__":
        if os.environ.get("FAKE_DIE_AT_START") == str(rank):
            sys.exit(7)
        out = dict(rank=rank, world=int(os.environ["WORLD_SIZE"]), transport="fake", pid=os.getpid())
    elif fn == "env":
        out = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                              "OMC_RDZV_NONCE", "HSA_ENABLE_IPC_MODE_LEGACY")}
        out["device"] = device
    elif fn == "bad":
        print(json.dumps(dict(id=rid, ok=False, type="ValueError", error="S0, K, T must be positive.")), flush=True)
        continue
    elif fn == "bad_on_one":  # a rank-LOCAL ValueError (a library check that fails for one shard only): the peers go ahead
        if rank == kw.get("rank"):
            print(json.dumps(dict(id=rid, ok=False, type="ValueError", error="selected another number of rows")), flush=True)
            continue
        time.sleep(300)  # ... into a collective the refusing rank never enters
        continue
    elif fn == "fail_then_hang":
        if rank == kw.get("rank"):
            print(json.dumps(dict(id=rid, ok=False, type="RuntimeError", error="kernel launch failed")), flush=True)
        time.sleep(300)  # the failing rank hangs in its teardown, the others inside a collective
        continue
    elif fn == "die" and rank == kw.get("rank"):
        sys.exit(5)
    elif fn == "sleep" and rank == kw.get("rank"):
        time.sleep(kw["seconds"])
        out = {}
    elif fn == "noise":
        print("a stray line from some library")  # not JSON: the parent must skip it
        out = dict(rank=rank)
    else:
        out = dict(rank=rank, echo=kw)
    print(json.dumps(dict(id=rid, ok=True, result=out)), flush=True)
