#!/usr/bin/env python3
"""Headline benchmark: whole American-option pricings per second on MI355X.

One "step" = one full pricing of the workload through the C ABI (libomc.so):
Philox/Box-Muller path generation into the [step][path] float32 matrix -> polynomial
Longstaff-Schwartz backward induction -> discounted mean.  Nothing is cached between
steps (each step uses a fresh Philox sub-stream); inputs are scalars, so there is no
host->device data and `value` is the HBM-resident rate by construction.

Workload (BASELINE.json configs[1]): GBM American put, S0=K=100, r=5%, sigma=20%, T=1,
1,000,000 paths x 252 steps per GPU, polynomial LSM.  With N GPUs every rank prices its
own 1M-path shard of an N x 1M-path problem (weak scaling; BASELINE configs[2] is the
same thing at 8M paths per GPU: --paths-per-gpu 8000000), moments and sums all-reduced
over RCCL.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline          the dominant kernel (largest time share): algorithmic bytes per launch / mean
                    HIP-event duration of that launch inside the timed region, vs 8 TB/s HBM peak
  roofline_pathgen  the path-generation kernel named by north_star: (n_steps+1)*n_paths*4 bytes
  roofline_kernels  every big kernel of the pricing, same accounting (+ PMC traffic if profiled)
  cpu_baseline  the C oracle (oracle/, a port: the reference is Python) on this host
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


def lsm_algorithmic_bytes(semantics: str, M: int, N: int) -> float:
    """DESIGN.md 'Algorithmic bytes': what the backward induction must move per pricing."""
    if semantics == "two_pass":
        # pass 1 reads S rows 1..N-1 once (+ terminal row), pass 2 reads them again
        return 2.0 * 4 * M * N
    # per-step sweep t: S_t, S_{t-1} (4+4), state sx/tex (4+4) per path
    return 16.0 * M * (N - 1) + 12.0 * M


def cpu_baseline(M, N, semantics, budget_s=25.0):
    """Oracle port timed on this host, bounded sample of the same workload."""
    from oracle import cpu as orc
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
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    # size the sample so the CPU leg stays ~10-30 s: probe with 1/50 of the paths
    probe = max(2000, (M // 50) // 4 * 4)
    t0 = time.perf_counter()
    S = orc.gbm_paths(probe, N, 100.0, 0.05, 0.2, 1.0, 42)
    orc.lsm_poly(S, 100.0, 0.05, 1.0, True, semantics)
    dt = time.perf_counter() - t0
    frac = min(1.0, budget_s / max(dt * M / probe, 1e-9))
    Ms = max(probe, int(M * frac) // 4 * 4)
    t0 = time.perf_counter()
    S = orc.gbm_paths(Ms, N, 100.0, 0.05, 0.2, 1.0, 42)
    t1 = time.perf_counter()
    res = orc.lsm_poly(S, 100.0, 0.05, 1.0, True, semantics)
    t2 = time.perf_counter()
    return {
        "value": Ms * N / (t2 - t0), "unit": "path-steps/s", "cores": threads, "kind": "port",
        "sample": f"{Ms} paths x {N} steps, same workload/semantics; path-gen {t1 - t0:.2f}s "
                  f"+ LSM sweeps {t2 - t1:.2f}s, both OpenMP x{threads}; C oracle, f32 paths/f64 sums",
        "price": res["price"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--n-steps", type=int, default=252)
    ap.add_argument("--semantics", default="two_pass", choices=["two_pass", "reference", "textbook"])
    ap.add_argument("--model", default="gbm", choices=["gbm", "heston"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--group", type=int, default=10,
                    help="pricings enqueued per omc_price_american_seq call (one host wait per group)")
    ap.add_argument("--sync-every-step", action="store_true",
                    help="one synchronous omc_price_american call per step instead")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --single-device rehearses world_size > 1 on one GPU")
    ap.add_argument("--single-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="go through torch.distributed/RCCL even with one rank (rehearsal of the N>1 path)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    M, N = a.paths_per_gpu, a.n_steps

    import torch
    from options_model_amd import _ffi, dist as omc_dist

    if not torch.cuda.is_available() or _ffi.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist_mode = world > 1 or a.force_dist
    stdout_fd = None
    if dist_mode:
        # RCCL prints a version banner on stdout when its first communicator comes up; the contract
        # is ONE JSON line on stdout, so everything before it goes to stderr at the descriptor level
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if a.backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group("gloo", rank=rank, world_size=world)
        pricer = omc_dist.ShardedPricer(local_rank, force_hook=a.force_dist)
        barrier = td.barrier
    else:
        pricer = None
        ctx = _ffi.Context(local_rank)
        barrier = lambda: None  # noqa: E731

    kw = dict(model=a.model, is_put=(a.model == "gbm"), semantics=a.semantics, n_steps=N, seed=42)

    def one_step(i):
        if pricer is not None:
            out = pricer.price_american(M * world, stream=i, **kw)
            return out, out["local"]
        out = ctx.price_american(_ffi.make_params(n_paths=M, stream=i, **kw))
        return out, out

    for i in range(a.warmup):
        one_step(1000 + i)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_paths = ms_lsm = ms_p1 = ms_p2 = 0.0
    price = 0.0
    if not a.sync_every_step:
        # The K pricings are enqueued through omc_price_american_seq in groups of `--group` (no host
        # synchronisation inside a group: pricing i + 1 is launched while pricing i runs; with several
        # ranks the all-reduces are stream-ordered too); every group's first pricing carries the HIP
        # events the per-kernel times come from.
        nsamp = 0
        for lo in range(0, a.steps, a.group):
            ids = list(range(lo, min(lo + a.group, a.steps)))
            if pricer is not None:
                louts = pricer.price_american_seq(M * world, ids, **kw)
                outs = [o["local"] for o in louts]
                price = louts[-1]["price"]
            else:
                outs = ctx.price_american_seq([_ffi.make_params(n_paths=M, stream=i, **kw) for i in ids])
                price = outs[-1]["price"]
            ms_paths += outs[0]["ms_paths"]
            ms_lsm += outs[0]["ms_lsm"] if len(outs) == 1 else sum(o["ms_total"] for o in outs) / len(outs) - outs[0]["ms_paths"]
            ms_p1 += outs[0].get("ms_pass1", 0.0)
            ms_p2 += outs[0].get("ms_pass2", 0.0)
            nsamp += 1
        scale = a.steps / nsamp  # the averages below divide by a.steps
        ms_paths *= scale; ms_lsm *= scale; ms_p1 *= scale; ms_p2 *= scale
    else:
        for i in range(a.steps):
            out, loc = one_step(i)
            ms_paths += loc["ms_paths"]
            ms_lsm += loc["ms_lsm"]
            ms_p1 += loc.get("ms_pass1", 0.0)
            ms_p2 += loc.get("ms_pass2", 0.0)
            price = out["price"]
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist_mode:
        import torch.distributed as td
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_paths /= a.steps
    ms_lsm /= a.steps
    ms_p1 /= a.steps
    ms_p2 /= a.steps
    b_gen = 4.0 * (N + 1) * M
    b_lsm = lsm_algorithmic_bytes(a.semantics, M, N)
    line = {
        "metric": "paths x steps / sec (whole American pricing: path-gen + LSM + mean)",
        "value": world * M * N * a.steps / elapsed,
        "unit": "path-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.model.upper()} American {'put' if a.model == 'gbm' else 'call'}, "
                               f"S0=K=100 r=0.05 sigma=0.2 T=1, {M} paths x {N} steps per GPU, "
                               f"polynomial LSM [1,u,u^2] ({a.semantics} flow)",
                   "paths_per_gpu": M, "n_steps": N, "semantics": a.semantics,
                   "parallelism": f"path-sharded x{world}" if world > 1 else "single GPU",
                   "rng": "Philox4x32-10 + Box-Muller, antithetic"},
        "paths_x252_per_sec_per_gpu": M * N * a.steps / elapsed / 252.0,
        "price": price,
    }

    # ---- roofline: every big kernel of the pricing, HIP-event time per launch inside the timed
    # region (library events on its own stream) vs its algorithmic bytes; `roofline` is the
    # dominant one (largest time share), `roofline_pathgen` the one north_star names.
    def rf(kernel, nbytes, ms, launches=1):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": nbytes, "ms_per_launch": ms,
                "launches_per_pricing": launches, "traffic": None}

    kernels = [rf(f"{a.model}_paths_kernel", b_gen, ms_paths)]
    if a.semantics == "two_pass":
        kernels.append(rf("lsm_pass1_kernel", 4.0 * M * N, ms_p1))  # rows 1..N-1 + terminal row
        kernels.append(rf("lsm_pass2_kernel", 4.0 * M * N, ms_p2))  # rows N..1 (upper bound)
    else:
        kernels.append(rf("lsm_step_kernel", 16.0 * M, ms_lsm / N, launches=N))  # S_t, S_t-1, sx, tex
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):  # measured HBM bytes per launch (rocprofv3 --pmc passes, see profiles/)
        try:
            pk = json.load(open(prof)).get("kernels", {})
            for k in kernels:
                stem = k["kernel"].replace("gbm_", "").replace("heston_", "")
                for name, v in pk.items():
                    if stem in name:
                        k["traffic"] = v["read_bytes"] + v["write_bytes"]
        except Exception:
            pass
    dominant = max(kernels, key=lambda k: k["ms_per_launch"] * k["launches_per_pricing"])
    line["roofline"] = dominant
    line["roofline_pathgen"] = kernels[0]
    line["roofline_kernels"] = kernels
    line["roofline_lsm_total"] = {"bytes_per_pricing": b_lsm, "ms_per_pricing": ms_lsm,
                                  "achieved": b_lsm / (ms_lsm * 1e-3) / 1e9, "unit": "GB/s"}

    if not dist_mode and not a.no_variants:
        var = {}
        for sem in ("two_pass", "reference", "textbook"):
            if sem == a.semantics:
                continue
            k2 = dict(kw, semantics=sem)
            ctx.price_american(_ffi.make_params(n_paths=M, stream=77, **k2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = max(3, a.steps // 4)
            for i in range(reps):
                o = ctx.price_american(_ffi.make_params(n_paths=M, stream=i, **k2))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            var[sem] = {"path_steps_per_s": M * N / dt, "ms_per_pricing": 1e3 * dt, "ms_lsm": o["ms_lsm"],
                        "price": o["price"]}
        if a.model == "gbm":
            # BASELINE config 5: the NN regressor (7->64->64->1) on the same paths x steps; the
            # network is trained by the library's fused float32-MFMA kernels (omc_mlp_train_epoch)
            from options_model_amd import nn_regressor as nnr
            nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, seed=1, nn_epochs=2)  # warm
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            o = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, M, N, seed=42)
            dt = time.perf_counter() - t1
            tk = o.timings_ms.get("seconds_train_kernels", 0.0) * 1e-3
            flop = 2 * (8 * 64 + 64 * 64 + 64) + 2 * 2 * 64 * 64 + 2 * 8 * 64  # per row: fwd, dH1, gW2, gW1
            rows_seen = o.sum_nitm * o.info.get("epochs_run", 0)
            var["nn_2x64"] = {"path_steps_per_s": M * N / dt, "seconds": dt, "price": o.price,
                              "rows": o.sum_nitm, "train_kernel_seconds": tk, "info": o.info,
                              "timings_ms": {k: round(v, 3) for k, v in o.timings_ms.items()}}
            if rows_seen and tk > 0:
                var["nn_2x64"]["train_mfma"] = {"bound": "mfma", "achieved": rows_seen * flop / tk / 1e12,
                                                "peak": 157.3, "unit": "TFLOP/s",
                                                "frac": rows_seen * flop / tk / 1e12 / 157.3}
        line["variants"] = var

    if rank == 0 and not dist_mode and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(M, N, a.semantics)
    elif rank == 0:
        line["cpu_baseline"] = None
    if stdout_fd is not None:
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist_mode:
        import torch.distributed as td
        pricer.close()
        td.destroy_process_group()
    else:
        ctx.close()


if __name__ == "__main__":
    main()
