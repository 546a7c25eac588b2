#!/usr/bin/env python3
"""Headline benchmark: whole American-option pricings per second on MI355X.

One "step" = one full pricing of the workload through the C ABI (libomc.so):
Philox/Box-Muller path generation into the [step][path] float32 matrix -> polynomial
Longstaff-Schwartz backward induction -> discounted mean.  Nothing is cached between
steps (each step uses a fresh Philox sub-stream); inputs are scalars, so there is no
host->device data and `value` is the HBM-resident rate by construction.

Storage of the paths (`--storage`, `config.storage` in the line): the library's default for antithetic GBM in the two-pass
flow is FOLDED -- only the first partner of every antithetic pair is generated and stored ([step][path/2]), the partner is
priced from the same spot (S_t S'_t = S0^2 exp(2 drift t); include/omc.h, option "fold_antithetic") -- so every kernel's
algorithmic bytes are half those of the full matrix, and the line says so; `--storage full` runs the two-halves matrix of
rounds 1-5 (and of Heston, which cannot fold).

Workload (BASELINE.json configs[1], `--config c2`): GBM American put, S0=K=100, r=5%, sigma=20%,
T=1, 1,000,000 paths x 252 steps per GPU, polynomial LSM.  `--config c3` is BASELINE configs[2]'s
per-GPU shard (8,000,000 paths per GPU: 64M paths over 8 GPUs), `--config c4` configs[3] (Heston
call, 4,000,000 paths).  With N GPUs every rank prices its own shard of an N x paths-per-GPU problem
(weak scaling), regression moments and payoff sums all-reduced over RCCL.

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts N rank processes itself,
BEFORE this process makes any GPU call; under `python -m torch.distributed.run` the ranks are the
launcher's.  Either way: world size must equal --gpus, the node must have that many devices, and
the JSON carries the communicator's own rank count (`rccl_ranks`).

Every default line (`--config c2`, two-pass flow; N = 1 included) also carries a `config3` block: BASELINE configs[2] ITSELF
-- 64M paths x 252 steps in total, sharded over the job's N ranks (8M per GPU at N = 8), strong scaling -- timed through the
job's communicator, with every rank's communicator-free time on its own shard and rank 0's time for the whole problem on
one card beside it: `scaling_efficiency`, `speedup_vs_one_gpu`, and the sharded price checked against the one-GPU price
(config3_block below; `--no-config3` / `--config3-paths` / `--config3-steps`).  The headline stays configs[1] per GPU.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline          the dominant kernel (largest time share): algorithmic bytes per launch / mean
                    HIP-event duration of that launch inside the timed region, vs 8 TB/s HBM peak
  roofline_pathgen  the path-generation kernel named by north_star: (n_steps+1)*n_paths*4 bytes
  roofline_kernels  every big kernel of the pricing, same accounting (+ PMC traffic, source labelled)
  roofline_per_step the per-timestep kernel north_star specifies (reference per-step flow), always
  price_check       GPU vs CPU oracle on the SAME Philox (seed, stream), bounded slice
  sustained         >= 3 s of back-to-back pricings (steady clocks), same kernels
  cpu_baseline      the C oracle (oracle/, a port: the reference is Python) on this host
  config3           BASELINE configs[2] on this job's ranks (see above)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable

CONFIGS = {  # BASELINE.json configs[1..3]
    "c2": dict(model="gbm", paths_per_gpu=1_000_000),
    "c3": dict(model="gbm", paths_per_gpu=8_000_000),
    "c3x1": dict(model="gbm", paths_per_gpu=64_000_000),  # ALL of configs[2] on one GPU (65 GB of paths)
    "c4": dict(model="heston", paths_per_gpu=4_000_000),
    "c5": dict(model="gbm", paths_per_gpu=1_000_000),  # configs[4]: the NN regressor (2 x 64): its own flow, bench_c5()
    "c1nn": dict(model="gbm", paths_per_gpu=10_000),   # configs[0] as the reference itself runs it: its default call, bench_c1nn()
}
MARKET = dict(S0=100.0, K=100.0, r=0.05, sigma=0.2, T=1.0)
HESTON = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)


def step_bytes_per_path(semantics: str) -> float:
    """what one launch of lsm_step_kernel must move per path."""
    # reference: S_t, S_t-1 and the path's `live` value (S_N, or negative once exercised): 4 bytes each;
    # textbook: S_t, S_t-1, sx, tex
    return 12.0 if semantics == "reference" else 16.0


def lsm_algorithmic_bytes(semantics: str, M: int, N: int) -> float:
    """DESIGN.md 'Algorithmic bytes': what the backward induction must move per pricing."""
    if semantics == "two_pass":
        # pass 1 reads S rows 1..N-1 once (+ terminal row), pass 2 reads them again
        return 2.0 * 4 * M * N
    return step_bytes_per_path(semantics) * M * (N - 1) + 12.0 * M


def cpu_threads() -> int:
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) is the real core count
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                threads = max(1, min(threads, int(float(quota) / float(period) + 0.5)))
            break
        except Exception:
            continue
    return threads


def oracle_price(model, M, N, semantics, is_put, seed, stream):
    """The CPU oracle on the Philox (seed, stream) the GPU used -> (result dict, t_paths, t_lsm)."""
    from oracle import cpu as orc
    t0 = time.perf_counter()
    if model == "gbm":
        S = orc.gbm_paths(M, N, MARKET["S0"], MARKET["r"], MARKET["sigma"], MARKET["T"], seed, stream)
    else:
        S = orc.heston_paths(M, N, MARKET["S0"], MARKET["r"], MARKET["T"], HESTON["v0"], HESTON["kappa"],
                             HESTON["theta"], HESTON["xi"], HESTON["rho"], seed, stream, 0, 1)  # full truncation
    t1 = time.perf_counter()
    res = orc.lsm_poly(S, MARKET["K"], MARKET["r"], MARKET["T"], is_put, semantics)
    t2 = time.perf_counter()
    return res, t1 - t0, t2 - t1


def cpu_baseline(model, M, N, semantics, is_put, seed, stream, budget_s=25.0):
    """Oracle port timed on this host, bounded sample of the same workload (same seed / stream)."""
    threads = cpu_threads()
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    # size the sample so the CPU leg stays ~10-30 s: probe with 1/50 of the paths
    probe = max(2000, (M // 50) // 4 * 4)
    _, ta, tb = oracle_price(model, probe, N, semantics, is_put, seed, stream)
    frac = min(1.0, budget_s / max((ta + tb) * M / probe, 1e-9))
    Ms = max(probe, int(M * frac) // 4 * 4)
    res, ta, tb = oracle_price(model, Ms, N, semantics, is_put, seed, stream)
    return {
        "value": Ms * N / (ta + tb), "unit": "path-steps/s", "cores": threads, "kind": "port",
        "sample": f"{Ms} paths x {N} steps (first {Ms} paths of Philox seed {seed} stream {stream}), same "
                  f"workload/semantics; path-gen {ta:.2f}s + LSM sweeps {tb:.2f}s, both OpenMP x{threads}; "
                  f"C oracle, f32 paths/f64 sums",
        "price": res["price"],
    }


# ---------------------------------------------------------------------------------------------
def launch_ranks(n: int, deadline_s: float, argv=None) -> int:
    """Parent of a `--gpus N` run without a launcher: start N rank processes (this script again, with
    the rank environment) and relay rank 0's JSON line -- options_model_amd.launcher.launch_ranks, the machinery
    the facade's `n_gpus = N` uses as well: no GPU call here, children are never re-exec'd or retried, a failing
    child or the overall deadline (`--rank-timeout`) ends the job non-zero with the ranks named on stderr."""
    from options_model_amd import launcher
    return launcher.launch_ranks(n, deadline_s, argv or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                 who="bench.py", timeout_flag="--rank-timeout")


def bench_c5(a) -> int:
    """BASELINE configs[4] as a bench line of its own: GBM American put, 1M paths x 252 steps, NN continuation-value
    regressor SingleLSMNet(7, 64, 2) -- rows + normalisers, 25 epochs of the float32-MFMA trainer, the sticky pass 2,
    all in the library's kernels.  A step is one whole pricing.  The dominant kernel is mlp_train_kernel: its roofline is
    the float32 matrix-core peak, `achieved` = algorithmic FLOP of the rows trained / the time spent in the training
    launches (the rocprofv3 averages of the same run are in profiles/r04_c5_kernel_stats.csv)."""
    import torch
    from options_model_amd import nn_regressor as nnr
    M, N = a.paths_per_gpu or CONFIGS["c5"]["paths_per_gpu"], a.n_steps
    steps, warm = max(1, min(a.steps, 5)), max(1, min(a.warmup, 2))
    nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=1, nn_epochs=2)
    for i in range(warm):
        nnr.price_american_option_nn(MARKET["S0"], MARKET["K"], MARKET["r"], MARKET["sigma"], MARKET["T"], M, N, seed=7 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [nnr.price_american_option_nn(MARKET["S0"], MARKET["K"], MARKET["r"], MARKET["sigma"], MARKET["T"], M, N,
                                         seed=42 + i) for i in range(steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    o = outs[0]
    flop = 2 * (8 * 64 + 64 * 64 + 64) + 2 * 2 * 64 * 64 + 2 * 8 * 64  # per row: forward, dH1, gW2, gW1
    tk = sum(x.timings_ms.get("train_kernels", 0.0) for x in outs) * 1e-3
    rows = sum(x.sum_nitm * x.info.get("epochs_run", 0) for x in outs)
    tf = rows * flop / tk / 1e12 if tk > 0 else None
    line = {
        "metric": "paths x steps / sec (whole American pricing: path-gen + LSM + mean)", "value": M * N / dt,
        "unit": "path-steps/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * dt,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"GBM American put, S0=K=100 r=0.05 sigma=0.2 T=1, {M} paths x {N} steps, NN regressor "
                               f"SingleLSMNet(7, 64, 2), 25 epochs, batch {o.info.get('batch')} (two_pass flow)",
                   "baseline_config": "c5", "paths_per_gpu": M, "n_steps": N, "parallelism": "single GPU"},
        "roofline": {"kernel": "mlp_train_kernel", "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": tf / 157.3 if tf else None, "traffic": None, "flop_per_row": flop, "rows_trained": rows,
                     "kernel_seconds": tk, "note": "float32 products (the reference's precision); the chip holds 2.14 GHz "
                     "under this load, the peak is quoted at 2.4 GHz (DESIGN.md 6.3)"},
        "cpu_baseline": {"value": 3500.0, "unit": "path-steps/s", "cores": 8, "kind": "reference",
                         "sample": "BASELINE.md section 2: the reference's own NN pricer on 10k x 50 (132-153 s); its "
                                   "settings are infeasible at this size (1.1e7 optimizer steps of batch 256), not re-run here"},
        "price": o.price, "stderr": o.stderr, "rows": o.sum_nitm, "info": o.info,
        "timings_ms": {k: round(v, 3) for k, v in o.timings_ms.items()},
        "price_check": "tests/test_gpu_nn_full.py: rows / normalisers / frozen-net decisions vs the oracle on this config",
    }
    print(json.dumps(line))
    return 0


def bench_c1nn(a) -> int:
    """BASELINE configs[0] the way the reference runs it -- its DEFAULT call: AdvancedOptionPricer(K, r, sigma, 'put')
    .price_american_enhanced_lsm(100, 1, 10000, 50) with reference arguments (options_model_3.py:340-358, 565-613:
    SingleLSMNet 3 x 128, minibatch min(256, R), <= 25 epochs, Adam, dropout 0.1 in training AND in pass 2), through the
    drop-in class.  A step is one whole pricing (paths, rows, ~22,000 optimizer steps, pass 2).  Dominant kernel: the
    16-row-tile trainer; its roofline is the float32 matrix-core peak, and the fraction is tiny BY CONSTRUCTION -- a
    256-row minibatch is 16 tiles on a 256-CU chip and every step waits for two kernel boundaries and one round of
    parameter fetches (profiles/r05_default_call_timeline.txt holds the measured timeline of a step)."""
    import torch
    from options_model_amd import AdvancedOptionPricer, RNGManager
    M, N = a.paths_per_gpu or CONFIGS["c1nn"]["paths_per_gpu"], (50 if a.n_steps == 252 else a.n_steps)
    steps, warm = max(1, min(a.steps, 10)), max(1, min(a.warmup, 3))

    def one(seed):
        q = AdvancedOptionPricer(MARKET["K"], MARKET["r"], MARKET["sigma"], "put", RNGManager(seed))
        price = q.price_american_enhanced_lsm(MARKET["S0"], MARKET["T"], M, N)
        return price, q.last_result

    for i in range(warm):
        one(7 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [one(42 + i) for i in range(steps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    price, o = outs[0]
    H = 128
    flop = 2 * (8 * H + 2 * H * H + H) + 2 * 2 * H * H + 2 * 2 * H * H + 2 * 8 * H  # per row: forward, dH x 2, gW x 2, gW1
    tk = sum(x.get("seconds_train_kernels", 0.0) for _, x in outs)
    rows = sum(x.get("R", 0) * x.get("epochs_run", 0) for _, x in outs)
    nsteps = sum(x.get("optimizer_steps", 0) for _, x in outs)
    tf = rows * flop / tk / 1e12 if tk > 0 else None
    prices = [p_ for p_, _ in outs]
    line = {
        "metric": "paths x steps / sec (whole American pricing: path-gen + LSM + mean)", "value": M * N / dt,
        "unit": "path-steps/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * dt,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"GBM American put, S0=K=100 r=0.05 sigma=0.2 T=1, {M} paths x {N} steps, the reference's default "
                               f"call: NN regressor SingleLSMNet(7, 128, 3), minibatch {o.get('batch')}, <= 25 epochs, dropout 0.1 "
                               f"in training and in pass 2 (AdvancedOptionPricer.price_american_enhanced_lsm)",
                   "baseline_config": "c1 (configs[0]) with the reference's own regressor", "paths_per_gpu": M, "n_steps": N,
                   "parallelism": "single GPU"},
        "roofline": {"kernel": "mlp_train_q16_kernel<128,3>", "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": tf / 157.3 if tf else None, "traffic": None, "flop_per_row": flop, "rows_trained": rows,
                     "optimizer_steps": nsteps, "kernel_seconds": tk,
                     "us_per_optimizer_step": 1e6 * tk / nsteps if nsteps else None,
                     "note": "latency-bound by construction: 16 workgroups per minibatch, two dependent launches per optimizer "
                             "step; the step's measured timeline is profiles/r05_default_call_timeline.txt"},
        "cpu_baseline": {"value": M * N / 133.0, "unit": "path-steps/s", "cores": 8, "kind": "reference",
                         "sample": "BASELINE.md section 2: this very call through the reference itself (imported, torch on 8 "
                                   "threads): 133-147 s per pricing, measured in the build container; quoted, not re-run "
                                   "(the reference does not travel to the GPU box)"},
        "price": price, "prices": prices, "stderr": o.get("stderr"), "rows": o.get("R"),
        "info": {k: o.get(k) for k in ("trainer", "trainer_kernel", "pass2", "rows", "batch", "epochs_run", "optimizer_steps", "best_loss")},
        # (median over the timed pricings: the first one after a change of row count pays the row tensor's allocation)
        "timings_ms": {k[len("seconds_"):]: round(1e3 * sorted(x.get(k, 0.0) for _, x in outs)[len(outs) // 2], 3)
                       for k in o if k.startswith("seconds_")},
        "price_check": "tests/test_gpu_nn.py::test_config1_nn_end_to_end_band (the reference's own seed band: 6.81 - 7.29), "
                       "tests/test_gpu_dropout.py (trainer and pass 2 under the oracle's masks)",
    }
    print(json.dumps(line))
    return 0


def config3_block(a, _ffi, ctx, pricer, rank, world, local_rank, N, kw, barrier, comm, seq_overlap):
    """BASELINE configs[2]: GBM American put, 64M paths x 252 steps, path-sharded over the job's ranks (8M per GPU at 8).
    Collective: every rank runs every part.  Three timings of the SAME pricings (Philox streams 5000...):
      sharded      the job's communicator (moment + result all-reduces, seq_overlap as the headline uses it), max over ranks
      shard alone  every rank its own shard (same pair offset) through a communicator-free context on its card
      one GPU      rank 0 prices the whole 64M-path problem alone on its card (65 GB of paths), the others wait
    -> scaling_efficiency = mean(shard alone) / sharded  (what the exchanges and rank skew cost),
       speedup_vs_one_gpu = one GPU / sharded             (the north-star's ">= 7x at 8 GPUs", measured, not modelled),
       and the sharded price must equal the one-GPU price to 1e-12 (same Philox pairs, sums in another order)."""
    want = a.config3_paths or 64 * (a.paths_per_gpu or CONFIGS["c2"]["paths_per_gpu"])
    total = want // (8 * world) * (8 * world)
    if total <= 0:
        return {"error": f"--config3-paths {want} is smaller than 8 x {world} ranks"}
    M3 = total // world
    K3, W3 = max(1, a.config3_steps), 3
    ids_w, ids = [4900 + i for i in range(W3)], [5000 + i for i in range(K3)]
    kw3 = dict(kw, semantics="two_pass")

    def timed(run):
        run(ids_w)
        barrier(); ctx.sync()
        t = time.perf_counter()
        outs = run(ids)
        barrier(); ctx.sync()
        return time.perf_counter() - t, outs

    out = {"workload": f"GBM American put, S0=K=100 r=0.05 sigma=0.2 T=1, {total} paths x {N} steps in TOTAL, path-sharded over "
                       f"{world} rank(s) ({M3} per GPU), polynomial LSM (two_pass flow): BASELINE configs[2]",
           "baseline_config": "c3 (configs[2])", "total_paths": total, "paths_per_gpu": M3, "n_steps": N, "n_gpus": world,
           "steps": K3, "warmup": W3, "scaling": "strong", "unit": "path-steps/s", "storage": a.storage}
    if pricer is None:  # one rank: the one-GPU leg IS the job
        def run1(s):
            return ctx.price_american_seq([_ffi.make_params(n_paths=total, pair_offset=0, stream=i, **kw3) for i in s])
        dt, outs = timed(run1)
        out.update(ms_per_step=1e3 * dt / K3, value=total * N * K3 / dt, price=outs[-1]["price"],
                   one_gpu={"ms_per_step": 1e3 * dt / K3, "rank": 0, "price": outs[-1]["price"]},
                   shard_alone_ms=[1e3 * dt / K3], scaling_efficiency=1.0, speedup_vs_one_gpu=1.0,
                   note="one rank: the whole problem on one card; the N > 1 lines of a scaling run carry the sharded timing, "
                        "each rank's own shard without a communicator, and rank 0's one-GPU time of this very problem")
        return out
    # (a) sharded, through the job's communicator
    dt, louts = timed(lambda s: pricer.price_american_seq(total, s, **kw3))
    dt = pricer.allreduce_max(dt)
    last = louts[-1]
    out.update(ms_per_step=1e3 * dt / K3, value=total * N * K3 / dt, price=last["price"], comm=comm, seq_overlap=seq_overlap,
               last_pricing={k: last[k] for k in ("n_paths", "n_exercised", "n_zero", "sum_nitm")})
    # (b) every rank its own shard, no communicator (a second context on the same card; the pair offset keeps the paths)
    from options_model_amd import dist as omc_dist
    n_local, off = omc_dist.shard(total, world, rank)
    alone = _ffi.Context(local_rank)
    alone.set_option("fold_antithetic", 1 if a.storage == "folded" else 0)
    try:
        def run_alone(s):
            return alone.price_american_seq([_ffi.make_params(n_paths=n_local, pair_offset=off, stream=i, **kw3) for i in s])
        run_alone(ids_w)
        barrier(); alone.sync()
        t = time.perf_counter()
        run_alone(ids)
        alone.sync()
        mine = 1e3 * (time.perf_counter() - t) / K3
    finally:
        alone.close()
    vec = [0.0] * world
    vec[rank] = mine
    per_rank = pricer.allreduce_sum(vec)
    out["shard_alone_ms"] = per_rank
    out["shard_alone_ms_mean"] = sum(per_rank) / world
    out["shard_alone_ms_max"] = max(per_rank)
    out["scaling_efficiency"] = out["shard_alone_ms_mean"] / out["ms_per_step"]
    out["scaling_efficiency_definition"] = ("mean over ranks of the time a rank needs for ITS shard of the same pricings without a "
                                            "communicator / the job's time per pricing through the communicator (max over ranks)")
    # (c) the whole problem on ONE card (rank 0; 4 bytes x (N + 1) x 64M = 65 GB of paths): the measured denominator of the speed-up
    one = [0.0, 0.0, 0.0]  # [ms per step, price, ok]
    if rank == 0:
        try:
            solo = _ffi.Context(local_rank)
            solo.set_option("fold_antithetic", 1 if a.storage == "folded" else 0)
            try:
                def run_solo(s):
                    return solo.price_american_seq([_ffi.make_params(n_paths=total, pair_offset=0, stream=i, **kw3) for i in s])
                run_solo(ids_w[:2])
                solo.sync()
                k1 = max(1, min(K3, 5))
                t = time.perf_counter()
                so = run_solo(ids[K3 - k1:])
                solo.sync()
                one = [1e3 * (time.perf_counter() - t) / k1, so[-1]["price"], 1.0]
            finally:
                solo.close()
        except _ffi.OmcError as e:  # e.g. not enough free memory on rank 0's card: the block says so, the job goes on
            print(f"bench.py rank 0: config3 one-GPU leg skipped: {e}", file=sys.stderr)
    one = pricer.allreduce_sum(one)  # ranks 1.. contribute zeros: everybody learns rank 0's figures (and waits for them here)
    if one[2] == 1.0:
        out["one_gpu"] = {"ms_per_step": one[0], "rank": 0, "price": one[1]}
        out["speedup_vs_one_gpu"] = one[0] / out["ms_per_step"]
        out["efficiency_vs_one_gpu"] = out["speedup_vs_one_gpu"] / world
        out["price_rel_diff_vs_one_gpu"] = abs(out["price"] - one[1]) / abs(one[1])
        out["price_equals_one_gpu"] = bool(out["price_rel_diff_vs_one_gpu"] <= 1e-12)
    else:
        out["one_gpu"] = None
        out["speedup_vs_one_gpu"] = None
    return out


# ---------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed pricings first (the GPU's clocks take ~10-20 ms of work to come up: "
                         "profiles/r02f shows the same kernel 1.46 -> 0.93 ms over its first 13 launches)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="c2: GBM put 1M paths/GPU (default); c3: 8M paths/GPU (64M over 8); c4: Heston call 4M")
    ap.add_argument("--paths-per-gpu", type=int, default=None, help="override the config's paths per GPU")
    ap.add_argument("--n-steps", type=int, default=252)
    ap.add_argument("--semantics", default="two_pass", choices=["two_pass", "reference", "textbook"])
    ap.add_argument("--model", default=None, choices=["gbm", "heston"], help="override the config's model")
    ap.add_argument("--storage", default="folded", choices=["folded", "full"],
                    help="how the fused pricing keeps antithetic GBM paths in the two-pass flow: 'folded' (library default: "
                         "only the first partner of every pair is stored, the partner is priced from the same spot through "
                         "S_t S'_t = S0^2 exp(2 drift t)) or 'full' (both partners stored, the layout of rounds 1-5)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="omc_set_option on the pricing context before anything runs (experiments; repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-sustained", action="store_true")
    ap.add_argument("--only-timed", action="store_true",
                    help="warm-up + timed region only (for rocprofv3 --pmc passes: every dispatch of a kernel is then "
                         "the same workload): no per-step extra, sustained loop, price check, variants, CPU baseline")
    ap.add_argument("--group", type=int, default=None,
                    help="pricings enqueued per omc_price_american_seq call (one host wait -- and, across GPUs, one result "
                         "collective -- per group); default: all --steps in one group, at most 50")
    ap.add_argument("--sync-every-step", action="store_true",
                    help="one synchronous omc_price_american call per step instead")
    ap.add_argument("--backend", default=None, choices=["rccl", "nccl", "gloo"],
                    help="rccl (default): RCCL called from inside libomc.so, no torch; nccl: torch.distributed "
                         "over RCCL; gloo (default with --single-device): CPU-side rehearsal of world_size > 1")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="go through the communicator even with one rank (rehearsal of the N>1 path)")
    ap.add_argument("--p2p-exchange", action="store_true",
                    help="several ranks, native communicator: exchange the per-step flows' moments by direct writes into "
                         "every peer's mailbox (omc_p2p_*, SURVEY 5.8(b)) instead of an all-reduce per time step; checked "
                         "against the collective before use, on every rank")
    ap.add_argument("--rank-timeout", type=float, default=300.0,
                    help="multi-rank runs: overall deadline in seconds; ranks still running then are ended and the job "
                         "exits non-zero (launcher and, inside every rank, a watchdog)")
    ap.add_argument("--min-warmup-seconds", type=float, default=0.3,
                    help="after the --warmup pricings keep pricing until two consecutive groups differ by < 2 %% and at "
                         "least this much time has passed (the GPU's clocks come up over the first ~0.1-0.3 s of work)")
    ap.add_argument("--config3-paths", type=int, default=None,
                    help="total paths of the `config3` block (BASELINE configs[2]: 64M paths x 252 steps, path-sharded over "
                         "the --gpus ranks: 8M per GPU at 8); default 64 x the headline's paths per GPU (= 64M unless "
                         "--paths-per-gpu shrinks the run); rounded down to a multiple of 8 x ranks")
    ap.add_argument("--config3-steps", type=int, default=10, help="timed pricings of the config3 block (3 untimed first)")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the config3 block (default: every `--config c2 --semantics two_pass` line carries it)")
    ap.add_argument("--kernel-samples", type=int, default=4,
                    help="at least this many pricings of the timed region carry their own HIP events")
    a = ap.parse_args()
    if a.group is None:
        a.group = max(1, min(a.steps, 50))
    if a.only_timed:
        a.no_cpu_baseline = a.no_variants = a.no_sustained = a.no_config3 = True
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if a.config == "c1nn":
        if a.gpus != 1:
            raise SystemExit("bench.py --config c1nn is a single-GPU line")
        sys.exit(bench_c1nn(a))
    if a.config == "c5":
        if a.gpus != 1:
            raise SystemExit("bench.py --config c5 is a single-GPU line (the sharded NN flow: tools/time_nn_sharded.py)")
        sys.exit(bench_c5(a))

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, a.rank_timeout))  # nothing above touched the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(`python bench.py --gpus N`, or torch.distributed.run --nproc-per-node N ... --gpus N)")
    local_rank = 0 if a.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    cfg = CONFIGS[a.config]
    model = a.model or cfg["model"]
    M = a.paths_per_gpu or cfg["paths_per_gpu"]
    N = a.n_steps
    is_put = model == "gbm"
    backend = a.backend or ("gloo" if a.single_device and world > 1 else "rccl")

    # the CPU oracle's OpenMP runtime reads OMP_NUM_THREADS when the library is first loaded (price_check
    # comes before cpu_baseline): pin it to the container's CPU quota now, not to the 256 visible threads
    os.environ.setdefault("OMP_NUM_THREADS", str(cpu_threads()))

    if a.single_device and world > 1:
        # rehearsal: the ranks share ONE card, so they share its memory too -- the byte budget for the path matrices
        # that stay resident when per-step pricings share their launches is per card, not per rank
        os.environ.setdefault("OMC_SEQ_STEP_BYTES", str(64e9 / world))

    from options_model_amd import _ffi

    ndev = _ffi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if not a.single_device and ndev < world:
        raise SystemExit(f"bench.py: --gpus {world} but this node shows {ndev} HIP device(s)")

    dist_mode = world > 1 or a.force_dist
    stdout_fd = None
    pricer = None
    comm = "none"
    if dist_mode:
        # RCCL prints a version banner on stdout when its first communicator comes up; the contract
        # is ONE JSON line on stdout, so everything before it goes to stderr at the descriptor level
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        from options_model_amd import dist as omc_dist
        # a rank that gets stuck anywhere below (communicator bring-up, a collective whose peer died) ends itself
        watchdog = omc_dist.Watchdog(a.rank_timeout, f"bench.py rank {rank}")
        if backend == "rccl":
            try:
                pricer = omc_dist.RcclPricer(local_rank, rank, world, init_timeout_s=min(a.rank_timeout, 300.0))
                comm = "rccl-native (ncclAllReduce enqueued by libomc.so)"
            except omc_dist.RcclUnavailable as e:
                # raised on EVERY rank or on none (the ranks vote): all of them take the torch transport together
                print(f"bench.py rank {rank}: native RCCL unavailable for this job ({e}); using torch.distributed",
                      file=sys.stderr)
                backend = "gloo" if a.single_device else "nccl"  # (two ranks cannot share a device under RCCL)
            # anything else (a rank that never voted, a context that cannot be created) ends this rank non-zero
        if pricer is None:
            import torch
            import torch.distributed as td
            torch.cuda.set_device(local_rank)
            if backend == "nccl":
                td.init_process_group("nccl", rank=rank, world_size=world,
                                      device_id=torch.device("cuda", local_rank))
            else:
                td.init_process_group("gloo", rank=rank, world_size=world)
            pricer = omc_dist.ShardedPricer(local_rank, force_hook=a.force_dist)
            comm = f"torch.distributed {backend} (all-reduce hook)"
        ctx = pricer.ctx
        barrier = pricer.barrier
        rccl_ranks = pricer.comm_ranks()
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: communicator has {rccl_ranks} ranks, expected {world}")
    else:
        ctx = _ffi.Context(local_rank)
        barrier = lambda: None  # noqa: E731
        rccl_ranks = 0

    ctx.set_option("fold_antithetic", 1 if a.storage == "folded" else 0)
    for kv in a.option:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    def params(sem, stream, n_paths=M, pair_offset=0):
        return _ffi.make_params(model=model, is_put=is_put, semantics=sem, n_steps=N, seed=42, stream=stream,
                                n_paths=n_paths, pair_offset=pair_offset,
                                heston_scheme="full_truncation" if model == "heston" else "reference")

    kw = dict(model=model, is_put=is_put, semantics=a.semantics, n_steps=N, seed=42,
              heston_scheme="full_truncation" if model == "heston" else "reference")

    def price_group(ids, sem=a.semantics):
        """Enqueue the pricings `ids` (Philox stream ids) back to back -> (price of the last, local results)."""
        if pricer is not None:
            louts = pricer.price_american_seq(M * world, ids, **dict(kw, semantics=sem))
            return louts[-1]["price"], [o["local"] for o in louts]
        outs = ctx.price_american_seq([params(sem, i) for i in ids])
        return outs[-1]["price"], outs

    def one_step(i):
        if pricer is not None:
            out = pricer.price_american(M * world, stream=i, **kw)
            return out["price"], out["local"]
        out = ctx.price_american(params(a.semantics, i))
        return out["price"], out

    # Across GPUs a sequence overlaps each pricing's moment all-reduce with the next pricing's paths + pass 1
    # (library option "seq_overlap", default on for more than one rank).  Before anything is timed, every rank
    # prices the same three streams both ways; unless all ranks see identical bits the overlap is switched off
    # for the run -- and the line says which it was.
    seq_overlap = "n/a"
    if pricer is not None and comm.startswith("rccl-native") and a.semantics == "two_pass" and not a.sync_every_step:
        if world > 1 or os.environ.get("OMC_BENCH_SELFCHECK") == "1":  # (the variable: rehearsal with one rank)
            def bits(outs):
                return [(o["sum"], o["sumsq"], o["n_exercised"], o["n_zero"], o["sum_nitm"]) for o in outs]
            ctx.set_option("seq_overlap", 1)
            on = bits(price_group([2000, 2001, 2002])[1])
            ctx.set_option("seq_overlap", 0)
            off = bits(price_group([2000, 2001, 2002])[1])
            agree = pricer.allreduce_max(0.0 if on == off else 1.0) == 0.0
            ctx.set_option("seq_overlap", 1 if agree else 0)
            seq_overlap = "on" if agree else "off (self-check: overlapped != sequential)"
        else:
            seq_overlap = "off (one rank)"

    # W untimed steps through the same entry point as the timed ones (buffers, code paths warm) ...
    if a.sync_every_step:
        for i in range(a.warmup):
            one_step(1000 + i)
    else:
        for lo in range(0, a.warmup, a.group):
            price_group([1000 + i for i in range(lo, min(lo + a.group, a.warmup))])
    # ... then warm-up by TIME, whatever --warmup says: the clocks of a GPU that was idle come up over its first
    # ~0.1-0.3 s of work (the same kernel ran 1.46 -> 0.93 ms over its first 13 launches, profiles/r02f_c4_*), so a
    # count-based warm-up of a 0.55 ms pricing ends long before they have.  Groups of `--group` pricings until two
    # consecutive groups differ by < 2 % AND --min-warmup-seconds have passed (cap 5 s).  Across ranks the decision
    # is collective (max over ranks), so every rank runs the same number of groups.
    barrier()
    ctx.sync()
    tw0 = time.perf_counter()
    prev, settled, wgroups = None, False, 0
    while True:
        tg = time.perf_counter()
        if a.sync_every_step:
            for i in range(a.group):
                one_step(1500 + i)
        else:
            price_group([1500 + i for i in range(a.group)])
        ctx.sync()
        cur = time.perf_counter() - tg
        wgroups += 1
        close = prev is not None and abs(cur - prev) <= 0.02 * max(cur, prev)
        spent = time.perf_counter() - tw0
        stop = (close and spent >= a.min_warmup_seconds) or spent >= 5.0
        if dist_mode:  # all ranks leave the loop together
            stop = pricer.allreduce_max(0.0 if stop else 1.0) == 0.0
        if stop:
            settled = close
            break
        prev = cur
    warm_s = time.perf_counter() - tw0
    # per-kernel HIP events: not only on every group's first pricing but on every `stride`-th one, so that the timed
    # region yields at least --kernel-samples samples (an event costs a stream marker, hence not on every pricing)
    ev_stride = max(1, a.steps // max(1, a.kernel_samples))
    ctx.set_option("seq_event_stride", ev_stride)
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    ms_paths = ms_lsm = ms_p1 = ms_p2 = 0.0
    price = 0.0
    last_stream = a.steps - 1
    nsamp = 0
    if not a.sync_every_step:
        # The K pricings are enqueued through omc_price_american_seq in groups of `--group` (no host
        # synchronisation inside a group: pricing i + 1 is launched while pricing i runs; with several
        # ranks the all-reduces are stream-ordered too); the pricings marked `timed` carry HIP events (on
        # the library's own stream): the per-kernel times are averaged over them.
        tot_lsm = 0.0
        for lo in range(0, a.steps, a.group):
            ids = list(range(lo, min(lo + a.group, a.steps)))
            price, outs = price_group(ids)
            last_out = outs[-1]
            for o in outs:
                if o.get("timed"):
                    ms_paths += o["ms_paths"]
                    ms_p1 += o.get("ms_pass1", 0.0)
                    ms_p2 += o.get("ms_pass2", 0.0)
                    nsamp += 1
            tot_lsm += sum(o["ms_total"] for o in outs)
        ms_paths /= max(nsamp, 1); ms_p1 /= max(nsamp, 1); ms_p2 /= max(nsamp, 1)
        ms_lsm = tot_lsm / a.steps - ms_paths
    else:
        for i in range(a.steps):
            price, loc = one_step(i)
            last_out = loc
            ms_paths += loc["ms_paths"]
            ms_lsm += loc["ms_lsm"]
            ms_p1 += loc.get("ms_pass1", 0.0)
            ms_p2 += loc.get("ms_pass2", 0.0)
        nsamp = a.steps
        ms_paths /= a.steps; ms_lsm /= a.steps; ms_p1 /= a.steps; ms_p2 /= a.steps
    barrier()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    ranks_table = None
    if dist_mode:
        # who ran where: every rank's card (HIP ordinal, PCI bus id) and its OWN time per step, through the transport the
        # pricing used.  Two ranks on one card, or a communicator that does not connect `world` distinct ranks, end the
        # job on every rank (all hold the same table) -- unless this is the one-GPU rehearsal (--single-device).
        info = ctx.device_info()
        rows = omc_dist.gather_rank_table(pricer.allreduce_sum, rank, world, info["device"], info["pci_bus_id"],
                                          1e3 * elapsed / a.steps)
        try:
            ranks_table = omc_dist.check_rank_table(rows, world, allow_shared_device=a.single_device)
        except ValueError as e:
            raise SystemExit(f"bench.py rank {rank}: {e}")
        elapsed = pricer.allreduce_max(elapsed)
    ctx.set_option("seq_event_stride", 0)

    # folded storage (omc_result.folded): half the matrix is written and read -- the algorithmic bytes of every kernel halve
    folded = bool(last_out.get("folded"))
    cols = M // 2 if folded else M
    b_gen = 4.0 * (N + 1) * cols
    b_lsm = lsm_algorithmic_bytes(a.semantics, cols, N)
    line = {
        "metric": "paths x steps / sec (whole American pricing: path-gen + LSM + mean)",
        "value": world * M * N * a.steps / elapsed,
        "unit": "path-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{model.upper()} American {'put' if is_put else 'call'}, "
                               f"S0=K=100 r=0.05 sigma=0.2 T=1, {M} paths x {N} steps per GPU, "
                               f"polynomial LSM [1,u,u^2] ({a.semantics} flow)",
                   "baseline_config": a.config, "paths_per_gpu": M, "n_steps": N, "semantics": a.semantics,
                   "parallelism": f"path-sharded x{world}" if world > 1 else "single GPU",
                   "rng": "Philox4x32-10 + Box-Muller, antithetic",
                   "storage": ("antithetic-folded: [n_steps+1][n_paths/2] float32, first partner of every pair; the partner's "
                               "moneyness is (C_t/K)/S_t - 1 in float64 (include/omc.h, option fold_antithetic)") if folded
                              else "full: [n_steps+1][n_paths] float32, both partners of every pair",
                   "arithmetic": "f32 paths, f64 moments / solve / decisions / sums"},
        "paths_x252_per_sec_per_gpu": M * N * a.steps / elapsed / 252.0,
        "price": price, "price_stream": last_stream,
        "last_pricing": {k: last_out[k] for k in ("n_paths", "n_exercised", "n_zero", "sum_nitm", "sum", "sumsq")},
        "rccl_ranks": rccl_ranks, "comm": comm, "seq_overlap": seq_overlap,
        "ranks": ranks_table,
        "slowest_rank": max(ranks_table, key=lambda t: t["ms_per_step"])["rank"] if ranks_table else None,
        "distinct_gpus": len({t["pci_bus_id"] for t in ranks_table}) if ranks_table else 1,
        "clock_settled": bool(settled),
        "warmup_by_time": {"seconds": warm_s, "pricings": wgroups * a.group, "rule": "two consecutive groups within 2 %, "
                           f">= {a.min_warmup_seconds} s (cap 5 s), after the --warmup pricings"},
        "kernel_event_samples": nsamp,
    }

    # ---- roofline: every big kernel of the pricing, HIP-event time per launch inside the timed
    # region (library events on its own stream) vs its algorithmic bytes; `roofline` is the
    # dominant one (largest time share), `roofline_pathgen` the one north_star names.
    def rf(kernel, nbytes, ms, launches=1):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": nbytes, "ms_per_launch": ms,
                "launches_per_pricing": launches, "traffic": None, "traffic_source": None}

    kernels = [rf(f"{model}_paths_kernel", b_gen, ms_paths)]
    if a.semantics == "two_pass":
        kernels.append(rf("lsm_pass1_fold_kernel" if folded else "lsm_pass1_kernel", 4.0 * cols * N, ms_p1))  # rows 1..N-1 + terminal row
        kernels.append(rf("lsm_pass2_fold_kernel" if folded else "lsm_pass2_kernel", 4.0 * cols * N, ms_p2))  # rows N..1 (upper bound)
    else:
        # per-step flows: the pricings of a group share their launches (K per launch; ms_lsm is per pricing)
        k_eff = 1 if a.sync_every_step else ctx.seq_step_width([params(a.semantics, i) for i in range(min(a.group, a.steps))])
        kernels.append(rf("lsm_step_kernel" if k_eff == 1 else "lsm_step_multi_kernel",
                          step_bytes_per_path(a.semantics) * M * k_eff, ms_lsm * k_eff / N, launches=N))
        kernels[-1]["pricings_per_launch"] = k_eff
    # HBM bytes per launch are PMC counters: they exist only in a rocprofv3 --pmc pass (two separate
    # passes, FETCH_SIZE doubled per the gfx950 note), so an ordinary run quotes the committed summary
    prof = os.path.join(ROOT, "profiles", f"pmc_traffic_{a.config}.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            pk = pj.get("kernels", {})
            same = pj.get("config", "c2") == a.config and pj.get("paths_per_gpu", 1_000_000) == M
            for k in kernels:
                stem = k["kernel"].replace("gbm_", "").replace("heston_", "")
                for name, v in pk.items():
                    if "gbm_paths_kernel" in name and (", false>" in name) != folded:
                        continue  # (the folded pricing's generator writes the first partners only: half the bytes)
                    if stem in name.replace("_ind_", "_") and same and \
                            v.get("pricings_per_launch", 1) == k.get("pricings_per_launch", 1):
                        k["traffic"] = v["read_bytes"] + v["write_bytes"]
                        k["traffic_source"] = f"profiles/pmc_traffic_{a.config}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE " \
                                              f"passes, {pj.get('round', 'r01')}; not measured in this run)"
        except Exception:
            pass
    # what the SQ counters of the committed pass say about the vector pipe (the folded sweeps are bound by float64 issue,
    # not by bytes: DESIGN.md section 6.3) -- quoted, like the traffic, from profiles/
    sqf = os.path.join(ROOT, "profiles", f"pmc_sq_summary_r06_{a.config}.txt")
    if os.path.exists(sqf) and a.paths_per_gpu in (None, CONFIGS[a.config]["paths_per_gpu"]):
        try:
            cur = None
            for ln in open(sqf):
                if ln and not ln.startswith(" ") and ":" in ln:
                    cur = ln.split("<")[0].split(":")[0].strip()
                elif "VALU issuing" in ln and cur:
                    pct = float(ln.split("VALU issuing")[1].split("%")[0])
                    for k in kernels:
                        if k["kernel"] == cur or (cur.endswith("paths_kernel") and k["kernel"].endswith("paths_kernel")):
                            k["valu_issue_frac"] = pct / 100.0
                            k["valu_issue_source"] = f"profiles/{os.path.basename(sqf)} (rocprofv3 --pmc SQ passes; not measured in this run)"
        except Exception:
            pass
    if folded:
        for k in kernels:
            k["note"] = ("folded storage halves this kernel's algorithmic bytes; what binds it now is vector issue (Philox + "
                         "Box-Muller in the generator, float64 moments / fits in the sweeps: valu_issue_frac, DESIGN.md 6.3), so "
                         "`frac` of the HBM peak is the honest but no longer the binding ratio; the same kernels on full "
                         "storage (--storage full): 0.74 / 0.65 / 0.77 of the HBM peak")
    dominant = max(kernels, key=lambda k: k["ms_per_launch"] * k["launches_per_pricing"])
    line["roofline"] = dominant
    line["roofline_pathgen"] = kernels[0]
    line["roofline_kernels"] = kernels
    line["roofline_lsm_total"] = {"bytes_per_pricing": b_lsm, "ms_per_pricing": ms_lsm,
                                  "achieved": b_lsm / (ms_lsm * 1e-3) / 1e9 if ms_lsm > 0 else 0.0, "unit": "GB/s"}
    # the whole pricing against the HBM roofline: the per-kernel split above is skewed by the Infinity Cache
    # (the generator "finishes" with up to 256 MB of its rows still dirty on chip; their write-back lands in
    # the kernel that runs next -- DESIGN.md section 8.3), the total is not
    whole = (b_gen + b_lsm) / (elapsed / a.steps) / 1e9
    line["roofline_whole_pricing"] = {"bound": "hbm", "bytes_per_pricing": b_gen + b_lsm, "achieved": whole,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": whole / HBM_PEAK_GBS}
    if folded:
        # what the same pricing rate would need on full storage (the bytes rounds 1-5 moved): above 1.0 of the peak is
        # exactly what folding buys -- the work no longer scales with those bytes
        full_b = 4.0 * (N + 1) * M + lsm_algorithmic_bytes(a.semantics, M, N)
        line["roofline_whole_pricing"]["full_storage_equivalent"] = {
            "bytes_per_pricing": full_b, "achieved": full_b / (elapsed / a.steps) / 1e9, "unit": "GB/s",
            "frac": full_b / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS}

    # ---- the per-timestep kernel north_star specifies, as a first-class number: the reference per-step
    # flow on the same workload, HIP-event time of its N launches (boundaries included) / N
    if a.only_timed:
        line["roofline_per_step"] = kernels[1] if a.semantics == "reference" else None
    elif a.semantics != "reference":
        exchange = "n/a (one rank)" if not dist_mode else "collective (all-reduce of 8K doubles per time step)"
        if dist_mode and comm.startswith("rccl-native") and (a.p2p_exchange or os.environ.get("OMC_BENCH_P2P") == "1"):
            # direct peer writes instead of a collective per step: connect (collective decision), then every rank
            # prices one small sequence both ways; only if all ranks see the same price (1e-12: the collective may
            # add the ranks' contributions in another order) and no exchange timed out does the section use it
            good = False
            if pricer.enable_p2p():
                # BOTH pricings run on EVERY rank whatever happens to either: a rank that skipped the collective form
                # after an error would leave its peers inside those all-reduces (mismatched collectives = a hang
                # until the watchdog).  Errors are remembered, the ranks vote afterwards.
                ids = [3000, 3001, 3002, 3003]
                on = off = None
                for use in (1, 0):
                    ctx.set_option("p2p_exchange", use)
                    try:
                        got = [o["sum"] for o in price_group(ids, "reference")[1]]
                    except Exception as e:  # a timed-out exchange surfaces as an error of the call (on every rank)
                        print(f"bench.py rank {rank}: self-check pricing (direct exchange {'on' if use else 'off'}) "
                              f"failed: {e}", file=sys.stderr)
                        got = None
                    if use:
                        on = got
                    else:
                        off = got
                good = on is not None and off is not None and all(abs(x - y) <= 1e-12 * abs(y) for x, y in zip(on, off))
            good = pricer.allreduce_max(0.0 if good else 1.0) == 0.0
            ctx.set_option("p2p_exchange", 1 if good else 0)
            exchange = ("direct writes into every peer's mailbox (omc_p2p_*), one launch per time step" if good
                        else exchange + "; the direct exchange did not pass its self-check")
        # (a) ONE pricing per launch: the latency of a single pricing's time step; (b) K pricings of the sequence
        # per launch (omc_price_american_seq, option "seq_step_k"): the throughput of the same kernel when the
        # chip is filled.  Same kernel body, same bits per pricing (tests/test_gpu_step_multi.py).
        def per_step(k, reps):
            ctx.set_option("seq_step_k", k)
            plist = [params("reference", i) for i in range(max(k, 2))]
            k_eff = ctx.seq_step_width(plist)
            price_group(list(range(900, 900 + max(k, 2))), "reference")  # warm (workspaces)
            barrier(); ctx.sync()
            t1 = time.perf_counter()
            pr, outs = price_group(list(range(reps)), "reference")
            barrier(); ctx.sync()
            dt = time.perf_counter() - t1
            if dist_mode:
                dt = pricer.allreduce_max(dt)
            ms_sweep = sum(o["ms_total"] for o in outs) / len(outs) - outs[0]["ms_paths"]  # per pricing
            r = rf("lsm_step_kernel" if k_eff == 1 else "lsm_step_multi_kernel",
                   step_bytes_per_path("reference") * M * k_eff, ms_sweep * k_eff / N, launches=N)
            r.update(pricings_per_launch=k_eff, price=pr, price_stream=reps - 1,
                     last_pricing={q: outs[-1][q] for q in ("n_paths", "n_exercised", "n_zero", "sum_nitm")},
                     ms_per_pricing=1e3 * dt / reps, path_steps_per_s=world * M * N * reps / dt,
                     us_per_time_step_per_pricing=1e3 * ms_sweep / N)
            return r
        reps1 = max(5, a.steps // 2)
        one = per_step(1, reps1)
        by_k = {1: one}
        for k in (4, 8, 16):
            try:
                r = per_step(k, 2 * k)
            except _ffi.OmcError as e:  # (e.g. out of device memory for K resident matrices): keep what was measured
                print(f"bench.py rank {rank}: per-step flow with {k} pricings per launch skipped: {e}", file=sys.stderr)
                break
            if r["pricings_per_launch"] not in by_k:
                by_k[r["pricings_per_launch"]] = r
        ctx.set_option("seq_step_k", -1)
        if os.path.exists(prof):  # PMC bytes per launch of the per-step kernels from the committed summary (same config)
            try:
                pj = json.load(open(prof))
                if pj.get("config", "c2") == a.config and pj.get("paths_per_gpu", 1_000_000) == M:
                    for r in by_k.values():
                        for name, v in pj.get("kernels", {}).items():
                            if r["kernel"] + "<" in name and v.get("pricings_per_launch", 1) == r["pricings_per_launch"]:
                                r["traffic"] = v["read_bytes"] + v["write_bytes"]
                                r["traffic_source"] = f"profiles/pmc_traffic_{a.config}.json ({pj.get('round', '')}; not measured in this run)"
            except Exception:
                pass
        best = max(by_k.values(), key=lambda r: r["frac"])
        r = dict(best)
        r.update(flow="reference (per-step sticky flow: Options_model.py:108-157)", exchange_across_ranks=exchange,
                 note="ms_per_launch = HIP-event time of the whole N-launch sweep / N (kernel boundaries included); a launch "
                      "advances `pricings_per_launch` independent pricings of the sequence by one time step",
                 single_pricing={"us_per_time_step": one["us_per_time_step_per_pricing"], "frac": one["frac"],
                                 "ms_per_pricing": one["ms_per_pricing"], "path_steps_per_s": one["path_steps_per_s"]},
                 by_pricings_per_launch={str(k): {"frac": v["frac"], "achieved": v["achieved"], "ms_per_launch": v["ms_per_launch"],
                                                  "traffic": v.get("traffic"),
                                                  "ms_per_pricing": v["ms_per_pricing"], "path_steps_per_s": v["path_steps_per_s"]}
                                         for k, v in sorted(by_k.items())})
        line["roofline_per_step"] = r
    else:
        line["roofline_per_step"] = kernels[1]

    # ---- >= 3 s of back-to-back pricings with the same kernels: steady clocks, and long enough for an outside
    # observer sampling the card once a second or so (the driver's `gpu_busy`) to see it busy
    if not a.no_sustained:
        target_s = 3.0
        n = max(a.group, int(target_s / max(elapsed / a.steps, 1e-6)) + 1)
        n = min(n, 20000)
        barrier(); ctx.sync()
        t1 = time.perf_counter()
        done = 0
        while done < n:
            ids = list(range(2000 + done, 2000 + min(done + 50, n)))
            price_group(ids)
            done += len(ids)
        barrier(); ctx.sync()
        dt = time.perf_counter() - t1
        if dist_mode:
            dt = pricer.allreduce_max(dt)
        line["sustained"] = {"pricings": n, "seconds": dt, "ms_per_step": 1e3 * dt / n,
                             "value": world * M * N * n / dt, "unit": "path-steps/s",
                             "vs_timed_region": (world * M * N * n / dt) / line["value"]}
        line["timed_vs_sustained_ms"] = line["ms_per_step"] / line["sustained"]["ms_per_step"]

    # ---- BASELINE configs[2] on THIS job's ranks: 64M paths x 252 steps in total, path-sharded (strong scaling: the
    # problem is fixed, the shard is 64M / N).  The headline above stays configs[1] per GPU (weak scaling), so that the
    # N = 1 line of a scaling run equals the single-GPU bench; this block is what the >= 7x-at-8-GPUs target is stated on.
    if a.config == "c2" and a.semantics == "two_pass" and model == "gbm" and not a.no_config3 and not a.sync_every_step:
        try:
            line["config3"] = config3_block(a, _ffi, ctx, pricer, rank, world, local_rank, N, kw, barrier, comm, seq_overlap)
        except _ffi.OmcError as e:  # (on every rank alike, or the job ends at the watchdog: the calls inside are collective)
            line["config3"] = {"error": repr(e)}

    # ---- parity in the bench line: GPU vs the CPU oracle on the SAME Philox (seed, stream), bounded slice
    if rank == 0 and not a.only_timed:
        Mc = min(M, 200_000) // 4 * 4
        try:
            g = _ffi.Context(local_rank) if dist_mode else ctx  # unsharded, no communicator
            g.set_option("fold_antithetic", 1 if a.storage == "folded" else 0)
            gp = g.price_american(params(a.semantics, last_stream, n_paths=Mc))["price"]
            if g is not ctx:
                g.close()
            op = oracle_price(model, Mc, N, a.semantics, is_put, 42, last_stream)[0]["price"]
            fo = None
            if folded:  # the folded pricing's own oracle on the stored half (same decisions, 1e-9), beside the full-matrix one
                from oracle import cpu as orc
                c0, gg = orc.fold_constants(MARKET["S0"], MARKET["K"], MARKET["r"], MARKET["sigma"], MARKET["T"], N)
                half = orc.gbm_paths(Mc // 2, N, MARKET["S0"], MARKET["r"], MARKET["sigma"], MARKET["T"], 42, last_stream, 0, 0)
                fo = orc.lsm_two_pass_folded(half, MARKET["K"], MARKET["r"], MARKET["T"], is_put, c0, gg)["price"]
            line["price_check"] = {"gpu": gp, "oracle": op, "abs_err": abs(gp - op),
                                   "oracle_folded": fo, "rel_err_folded": None if fo is None else abs(gp - fo) / abs(fo),
                                   "rel_err": abs(gp - op) / abs(op), "tolerance_rel": 1e-3,
                                   "same_stream": True, "seed": 42, "stream": last_stream, "paths": Mc,
                                   "note": "oracle = oracle/omc_oracle.c (f64 restatement pinned to the reference's "
                                           "fixtures) on the identical Philox normals"}
        except Exception as e:  # never lose the throughput line to the checker
            line["price_check"] = {"error": repr(e)}

    if not dist_mode and not a.no_variants:
        var = {}
        for sem in ("two_pass", "reference", "textbook"):
            if sem == a.semantics:
                continue
            ctx.price_american(params(sem, 77))
            ctx.sync()
            t1 = time.perf_counter()
            reps = max(3, a.steps // 4)
            for i in range(reps):
                o = ctx.price_american(params(sem, i))
            ctx.sync()
            dt = (time.perf_counter() - t1) / reps
            var[sem] = {"path_steps_per_s": M * N / dt, "ms_per_pricing": 1e3 * dt, "ms_lsm": o["ms_lsm"],
                        "price": o["price"]}
        if model == "gbm" and a.config == "c2":
            # BASELINE config 5: the NN regressor (7->64->64->1) on the same paths x steps; the
            # network is trained by the library's fused float32-MFMA kernels (omc_mlp_train_epoch)
            try:
                import torch
                from options_model_amd import nn_regressor as nnr
                nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=1, nn_epochs=2)  # warm
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                o = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, M, N, seed=42)
                dt = time.perf_counter() - t1
                tk = o.timings_ms.get("train_kernels", 0.0) * 1e-3
                flop = 2 * (8 * 64 + 64 * 64 + 64) + 2 * 2 * 64 * 64 + 2 * 8 * 64  # per row: fwd, dH1, gW2, gW1
                rows_seen = o.sum_nitm * o.info.get("epochs_run", 0)
                var["nn_2x64"] = {"path_steps_per_s": M * N / dt, "seconds": dt, "price": o.price,
                                  "rows": o.sum_nitm, "train_kernel_seconds": tk, "info": o.info,
                                  "timings_ms": {k: round(v, 3) for k, v in o.timings_ms.items()}}
                if rows_seen and tk > 0:
                    var["nn_2x64"]["train_mfma"] = {"bound": "mfma", "achieved": rows_seen * flop / tk / 1e12,
                                                    "peak": 157.3, "unit": "TFLOP/s",
                                                    "frac": rows_seen * flop / tk / 1e12 / 157.3}
            except Exception as e:
                var["nn_2x64"] = {"error": repr(e)}
        # the regressor the reference's v1 / v2 pricers really use: a fresh ContNet per time step (omc_contnet.hip),
        # at the size those files default to and at this workload's size
        try:
            cn = {}
            for tag, (m_, n_) in (("10k_x_50", (10_000, 50)), ("workload", (M, N))):
                pp = _ffi.make_params(model=model, is_put=is_put, semantics="reference", n_paths=m_, n_steps=n_, seed=42,
                                      heston_scheme="full_truncation" if model == "heston" else "reference")
                ctx.price_american_contnet(pp, 32, 10, 1e-3, 42)  # warm
                t1 = time.perf_counter()
                reps = 5 if m_ <= 100_000 else 2
                for k in range(reps):
                    o = ctx.price_american_contnet(pp, 32, 10, 1e-3, 42)
                dt = (time.perf_counter() - t1) / reps
                cn[tag] = {"paths": m_, "steps": n_, "ms_per_pricing": dt * 1e3, "ms_per_time_step": dt * 1e3 / max(1, n_ - 1),
                           "path_steps_per_s": m_ * n_ / dt, "price": o["price"], "training_rows": o["sum_nitm"]}
            cn["parity"] = ("the reference's v1 / v2 regressor (fresh ContNet per step, Options_model.py:14-25,127-142) is UNSEEDED "
                            "there, so a price-for-price match does not exist: checked as <= 0.2 % flipped exercise decisions and "
                            "2e-3 on the price against a torch restatement started from the same initial nets, and within 1 % of "
                            "the reference's recorded prices (tests/test_gpu_contnet.py)")
            var["contnet_per_step"] = cn
        except Exception as e:
            var["contnet_per_step"] = {"error": repr(e)}
        line["variants"] = var

    if rank == 0 and a.gpus == 1 and not a.force_dist and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(model, M, N, a.semantics, is_put, 42, last_stream)
        # the reference itself cannot run on the GPU box (it only exists in the build container): its measured
        # figure is quoted from BASELINE.md section 2 beside the port's
        line["cpu_baseline"]["reference_quoted"] = {
            "value": 3.5e3, "unit": "path-steps/s", "cores": 8, "kind": "reference",
            "sample": "BASELINE.md section 2: options_model_3.AdvancedOptionPricer, configs[0] (GBM American put, 10k paths x 50 "
                      "steps, its NN regressor, 133-147 s on the 8-vCPU build container); quoted, not re-run here"}
    elif rank == 0:
        line["cpu_baseline"] = None
    if stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist_mode:
        barrier()
        pricer.close()
        if comm.startswith("torch"):
            import torch.distributed as td
            td.destroy_process_group()
        watchdog.cancel()
    else:
        ctx.close()


if __name__ == "__main__":
    main()
