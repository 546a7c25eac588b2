#!/usr/bin/env python3
"""Run the five BASELINE.json configs through the public API on one MI355X and print one JSON
line each (price, wall time, path-steps/s, kernel times).  Config 3 runs its per-GPU shard
(8M of the 64M paths); config 5 states its NN hyper-parameters.  Output is kept under
profiles/ as the round's per-config record."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from options_model_amd import _ffi, price_american_option  # noqa: E402

HP = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)


def timed(fn, reps):
    fn()  # warm-up (allocations, code object load)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return out, (time.perf_counter() - t0) / reps


def main():
    only = set(sys.argv[1:])
    ctx = _ffi.default_context(0)
    rows = []

    def poly(name, M, N, model="GBM", option_type="put", reps=5, **kw):
        if only and name not in only:
            return
        for sem in ("two_pass", "per_step", "textbook"):
            res, dt = timed(lambda: price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, M, N, model=model,
                                                          option_type=option_type, semantics=sem,
                                                          seed=42, ctx=ctx, **kw), reps)
            rows.append(dict(config=name, model=model, option=option_type, paths=M, steps=N, regressor="poly",
                             semantics=sem, price=res.price, stderr=res.stderr, seconds=dt,
                             path_steps_per_s=M * N / dt, ms_paths=res.timings_ms["paths"],
                             ms_lsm=res.timings_ms["lsm"],
                             pathgen_GBps=4.0 * (N + 1) * M / (res.timings_ms["paths"] * 1e-3) / 1e9))
            print(json.dumps(rows[-1]), flush=True)

    poly("C1", 10_000, 50, reps=20)
    poly("C2", 1_000_000, 252)
    poly("C3_shard", 8_000_000, 252, reps=3)
    poly("C4", 4_000_000, 252, model="Heston", option_type="call", heston_params=HP, reps=3)
    if not only or "C5" in only:
        from options_model_amd import nn_regressor
        trainers = [t for t in ("hip", "torch") if not only or not ({"hip", "torch"} & only) or t in only]
        # warm the process (torch import, HIP module load, allocator pools) on a small problem
        nn_regressor.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=1, nn_epochs=2)
        for trainer in trainers:
            t0 = time.perf_counter()
            res = nn_regressor.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 1_000_000, 252, seed=42,
                                                        nn_trainer=trainer)
            dt = time.perf_counter() - t0
            rows.append(dict(config="C5", paths=1_000_000, steps=252, trainer=trainer,
                             regressor="nn 2x64 (SingleLSMNet(7,64,2)), 25 epochs max, Adam 1e-3, batch auto",
                             price=res.price, stderr=res.stderr, seconds=dt,
                             path_steps_per_s=1_000_000 * 252 / dt, R=res.sum_nitm, timings_ms=res.timings_ms))
            print(json.dumps(rows[-1]), flush=True)
    if "C5r" in only:  # config 5 with the network the reference itself would build: SingleLSMNet(7, 128, 3)
        from options_model_amd import nn_regressor
        nn_regressor.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=1, nn_epochs=2,
                                              nn_hidden=128, nn_layers=3)
        for trainer in [t for t in ("hip", "torch") if not ({"hip", "torch"} & only) or t in only]:
            t0 = time.perf_counter()
            res = nn_regressor.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 1_000_000, 252, seed=42,
                                                        nn_hidden=128, nn_layers=3, nn_trainer=trainer)
            dt = time.perf_counter() - t0
            rows.append(dict(config="C5r", paths=1_000_000, steps=252, trainer=trainer,
                             regressor="nn 3x128 (SingleLSMNet(7,128,3), the reference's default shape), 25 epochs max",
                             price=res.price, stderr=res.stderr, seconds=dt, path_steps_per_s=1_000_000 * 252 / dt,
                             R=res.sum_nitm, timings_ms=res.timings_ms, info=res.info))
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
