#!/usr/bin/env python3
"""SURVEY row f-3 measured: ONE evaluation of the Heston calibrator's objective (heston_calibration.py:404-472 calls
HestonPricer.price_options_batch once per optimizer iteration; :283-312 simulates every distinct expiry -- 100,000 paths x
100 steps by default, :75-90 -- and averages every strike of it).

Workload: 60 quotes = 10 strikes x 6 expiries (30 d ... 2 y), S0 = 100, r = 3 %, the reference's default parameters.
Prints one JSON line:
  surface      HestonPricer.price_options_batch = omc_heston_price_surface: all expiries in one launch, all quotes in one
               more, one wait                                                        -> ms per evaluation, path-steps/s
  per_expiry   the reference's loop shape: one omc_heston_price_strikes call (2 launches + a wait) per expiry
  bit_equal    the two return the same bits (same Philox sub-streams, same summation order)
  cpu_baseline the numpy restatement of the reference's own loop (oracle.reference_flow; kind "port"), ONE expiry timed
               and scaled to six (about 20 s of CPU for all six); `reference_quoted` = the reference itself, measured in
               the build container (it does not travel): 22.1 s per evaluation on 8 vCPUs
Kernel times and the VALU-pipe share of the simulation kernel come from rocprofv3 (tools/gpu_r06.sh calib ->
profiles/r06_calibrator_kernel_stats.csv, r06_pmc_calibrator.txt), not from this script.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--paths", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--evals", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from options_model_amd import _ffi
    from options_model_amd.heston_pricer import HestonPricer

    cfg = SimpleNamespace(n_mc_paths=a.paths, n_time_steps=a.steps, seed=42)
    prm = SimpleNamespace(kappa=2.0, theta=0.04, sigma=0.3, rho=-0.7, v0=0.04)
    Ts = np.array([30 / 365, 60 / 365, 91 / 365, 0.5, 1.0, 2.0])
    T = np.repeat(Ts, 10)
    K = np.tile(np.linspace(80.0, 125.0, 10), 6)
    ctx = _ffi.default_context(0)
    out = {"workload": f"{len(K)} quotes over {len(Ts)} expiries, {a.paths} paths x {a.steps} steps per expiry, calibrator scheme "
                       f"(variance floored at 1e-8, arithmetic Euler; heston_calibration.py:223-255)",
           "path_steps_per_evaluation": len(Ts) * a.paths * a.steps, "evaluations_timed": a.evals}
    a_pr, b_pr = HestonPricer(cfg), HestonPricer(cfg)
    pa, pb = a_pr.price_options_batch(prm, 100.0, K, T, 0.03), b_pr.price_options_batch_per_expiry(prm, 100.0, K, T, 0.03)
    out["bit_equal"] = bool(np.array_equal(pa, pb))
    out["prices_first3"] = [float(x) for x in pa[:3]]
    for name, pr, fn in (("surface", a_pr, a_pr.price_options_batch), ("per_expiry", b_pr, b_pr.price_options_batch_per_expiry)):
        for _ in range(20):
            fn(prm, 100.0, K, T, 0.03)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(a.evals):
            fn(prm, 100.0, K, T, 0.03)
        ctx.sync()
        dt = (time.perf_counter() - t0) / a.evals
        out[name] = {"ms_per_evaluation": 1e3 * dt, "path_steps_per_s": out["path_steps_per_evaluation"] / dt,
                     "host_waits_per_evaluation": 1 if name == "surface" else len(Ts),
                     "launches_per_evaluation": 2 if name == "surface" else 2 * len(Ts)}
    out["speedup_surface_vs_per_expiry"] = out["per_expiry"]["ms_per_evaluation"] / out["surface"]["ms_per_evaluation"]
    if not a.no_cpu:
        from oracle import reference_flow as rf
        rng = np.random.default_rng(42)
        n_sim = a.paths // 2
        t0 = time.perf_counter()
        z1, z2 = rng.standard_normal((n_sim, a.steps)), rng.standard_normal((n_sim, a.steps))
        S, _ = rf.heston_calibrator_paths_from_normals(z1, z2, 100.0, 0.03, float(Ts[3]), prm.v0, prm.kappa, prm.theta, prm.sigma, prm.rho)
        p = rf.strike_prices(S[:, -1], K[:10], 0.03, float(Ts[3]))
        one = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": a.paths * a.steps / one, "unit": "path-steps/s", "cores": 1, "kind": "port",
                               "seconds_per_evaluation": one * len(Ts),
                               "sample": f"one expiry of the same evaluation ({a.paths} x {a.steps}, 10 strikes) through the numpy "
                                         f"restatement of the reference's loop (oracle.reference_flow.heston_calibrator_paths_from_normals, "
                                         f"[path][step] float64 arrays as the reference builds them): {one:.2f} s, x {len(Ts)} expiries",
                               "price_atm": float(p[4]),
                               "reference_quoted": {"seconds_per_evaluation": 22.07, "value": 2.72e6, "unit": "path-steps/s", "cores": 8,
                                                    "kind": "reference",
                                                    "sample": "the reference's own HestonPricer.price_options_batch on this very workload, "
                                                              "imported in the build container (8 vCPUs; numpy, effectively one thread): "
                                                              "22.07 s per evaluation; quoted, the reference does not travel"}}
        out["gpu_vs_cpu_port"] = out["cpu_baseline"]["seconds_per_evaluation"] / (1e-3 * out["surface"]["ms_per_evaluation"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
