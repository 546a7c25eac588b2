#!/usr/bin/env python3
"""The NN regressor sharded over W rank processes sharing this GPU (librccl stand-in, tests/rccl_standin): where does an
epoch's time go -- selecting / gathering the rank's rows of the epoch's permutation vs the training launches (the
stand-in's all-reduce is a host round trip, so the training figure is an upper bound of what RCCL would show).
usage: time_nn_sharded.py [world] [paths] [steps] [epochs]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rccl_standin
os.environ["OMC_RCCL_LIB"] = rccl_standin.build()
for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
    os.environ.pop(k, None)
from options_model_amd import launcher
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
E = int(sys.argv[4]) if len(sys.argv) > 4 else 3
pool = launcher.pool(W, [0] * W)
kw = dict(S0=100.0, K=100.0, r=0.05, sigma=0.2, T=1.0, n_paths=M, n_steps=N, model="GBM", option_type="put",
          heston_params=None, seed=42, stream=0, nn_hidden=64, nn_layers=2, nn_epochs=E)
pool.call_all("price_american_option_nn", dict(kw, n_paths=20000, n_steps=10, nn_epochs=1), timeout_s=600)  # warm
t0 = time.perf_counter()
res = pool.call_all("price_american_option_nn", kw, timeout_s=1200)
dt = time.perf_counter() - t0
i = res[0]["info"]
print(json.dumps(dict(world=W, paths=M, steps=N, epochs=i["epochs_run"], rows=res[0]["sum_nitm"], rows_local=i["rows_local"],
                      batch=i["batch"], optimizer_steps=i["optimizer_steps"], seconds=dt,
                      seconds_select_per_epoch=i["seconds_select"] / i["epochs_run"],
                      seconds_train_per_epoch=i["seconds_train_epochs"] / i["epochs_run"], price=res[0]["price"])))
launcher.close_pools()
