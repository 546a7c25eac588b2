"""BASELINE-size checks through size-independent properties (the oracle is too slow to rerun at
these sizes inside a test): martingale of the generators, shard-count invariance through the
real all-reduce hook, determinism, flow ordering and agreement between sizes."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HP = dict(v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7)


def _last_row(ctx, S, M, N):
    last = np.empty(M, np.float32)
    ctx.lib.omc_memcpy_d2h(ctx.handle, last.ctypes.data, S.ptr + 4 * M * N, 4 * M)
    return last.astype(np.float64)


def test_config4_heston_martingale_and_european_consistency(ctx):
    """4M x 252 Heston paths (config 4 size): E[S_T] = S0 e^{rT}; the stored-path terminal row
    and the register-only European kernel see the same paths (same Philox stream)."""
    from options_model_amd import _ffi
    M, N = 4_000_000, 252
    S = ctx.heston_paths(M, N, 100.0, 0.05, 1.0, seed=21, stream=1, **HP)
    x = _last_row(ctx, S, M, N)
    S.free()
    assert abs(x.mean() - 100 * math.exp(0.05)) < 5 * x.std() / math.sqrt(M)
    call = np.maximum(x - 100.0, 0) * math.exp(-0.05)
    eu = ctx.price_european(_ffi.make_params(model="heston", is_put=False, n_paths=M, n_steps=N, seed=21,
                                             stream=1, **HP))
    assert eu["price"] == pytest.approx(call.mean(), rel=1e-6)


def test_config2_flows_are_ordered_and_reproducible(ctx):
    """1M x 252 GBM put: textbook LSM sits at the binomial anchor (~6.09, low-biased), the
    reference's sticky flows sit where SURVEY F2-F4 says (per-step below, two-pass above),
    European < textbook; the same call twice is bit-identical."""
    from options_model_amd import _ffi
    M, N = 1_000_000, 252
    out = {}
    for sem in ("two_pass", "reference", "textbook"):
        p = _ffi.make_params(semantics=sem, n_paths=M, n_steps=N, seed=1234)
        a = ctx.price_american(p)
        b = ctx.price_american(p)
        assert a["price"] == b["price"] and a["sumsq"] == b["sumsq"] and a["n_exercised"] == b["n_exercised"]
        out[sem] = a
    eu = ctx.price_european(_ffi.make_params(n_paths=M, n_steps=N, seed=1234))
    assert 6.02 < out["textbook"]["price"] < 6.12
    assert eu["price"] < out["textbook"]["price"] < out["two_pass"]["price"]
    assert abs(eu["price"] - 5.573526022256971) < 0.03
    assert out["two_pass"]["sum_nitm"] > out["reference"]["sum_nitm"]  # sticky mask shrinks the sets
    # the 8M-path config-3 shard agrees with the 1M-path run within Monte-Carlo error
    big = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=8_000_000, n_steps=N, seed=99))
    se = math.sqrt(out["two_pass"]["std"] ** 2 / M + big["std"] ** 2 / 8e6)
    assert abs(big["price"] - out["two_pass"]["price"]) < 5 * se + 0.01


def test_config3_style_shard_invariance_at_full_row_length(ctx):
    """2 x 500k-path shards with the moment tables exchanged through the hook == one 1M run."""
    import torch

    from options_model_amd import _ffi
    from options_model_amd import dist as omc_dist
    from options_model_amd.dist import _DevPtr
    M, N = 1_000_000, 252
    stream = torch.cuda.Stream()
    c = _ffi.Context(0, stream=stream.cuda_stream)
    mom = {}

    def run(rank, mode):
        n, off = omc_dist.shard(M, 2, rank)

        def h(dptr, count):
            if count != 8 * (N + 1):
                return
            t = torch.as_tensor(_DevPtr(dptr, count), device="cuda")
            if mode == "capture":
                mom[rank] = t.clone()
            else:
                t.add_(mom[1 - rank])

        c.set_allreduce_hook(h)
        with torch.cuda.stream(stream):
            out = c.price_american(_ffi.make_params(semantics="two_pass", n_paths=n, n_steps=N, seed=5,
                                                    pair_offset=off))
        c.set_allreduce_hook(None)
        return out

    run(0, "capture"), run(1, "capture")
    a, b = run(0, "inject"), run(1, "inject")
    c.close()
    full = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=5))
    assert (a["sum"] + b["sum"]) / M == pytest.approx(full["price"], rel=1e-12)
    assert a["n_exercised"] + b["n_exercised"] == full["n_exercised"]


def test_config4_size_heston_put_real_decisions_and_idempotence(ctx):
    """4M x 252 Heston PUT (config 4's size with the payoff that actually exercises early -- an American
    call on a non-dividend asset never does, so config 4 itself tests no decision).  Size-independent
    properties: (1) every flow exercises a large share of paths and prices above the European value on
    the same paths; (2) idempotence: replaying the two-pass fits through omc_lsm_apply_frozen returns the
    identical state and sums; (3) the per-step flow through the captured graph and kernel by kernel
    agree bit for bit; (4) the adapted (textbook) rule prices below the reference's look-ahead flows."""
    from options_model_amd import _ffi
    M, N = 4_000_000, 252
    S = ctx.heston_paths(M, N, 100.0, 0.05, 1.0, seed=77, stream=3, scheme=1, **HP)
    x = _last_row(ctx, S, M, N)
    european = (np.maximum(100.0 - x, 0) * math.exp(-0.05)).mean()
    tp = ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass", want_state=True)
    assert tp["n_exercised"] > 0.3 * M and tp["price"] > european
    b4 = np.concatenate([tp["betas"], tp["nitm"][:, None].astype(np.float64)], axis=1)
    rep = ctx.lsm_apply_frozen(S, 100.0, 0.05, 1.0, True, b4)
    assert np.array_equal(rep["tex"], tp["tex"]) and np.array_equal(rep["sx"], tp["sx"])
    assert rep["price"] == tp["price"] and rep["n_exercised"] == tp["n_exercised"]
    del rep
    ctx.set_option("step_graph", 1)
    a = ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "reference", want_state=True)
    ctx.set_option("step_graph", 0)
    b = ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "reference", want_state=True)
    ctx.set_option("step_graph", -1)
    assert a["price"] == b["price"] and np.array_equal(a["tex"], b["tex"]) and np.array_equal(a["sx"], b["sx"])
    assert a["n_exercised"] > 0.3 * M and a["price"] > european
    tb = ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "textbook")
    assert european < tb["price"] < a["price"] + 0.05  # textbook (adapted rule) below the look-ahead flows
    S.free()


def test_config4_as_baseline_words_it_heston_call_full_truncation_4m_x_252(ctx):
    """BASELINE configs[3] verbatim: Heston American CALL, full-truncation Euler, 4M paths x 252 steps, one GPU.
    (1) against the CPU oracle on the same Philox stream for the first 200k paths of the SAME global pair indices
    (1e-3 relative, north_star's tolerance; the oracle is float64 downstream of the same float32 paths);
    (2) at full size: exact homogeneity in (S0, K), the American call on a non-dividend asset worth at least the
    European one on the same paths, a sequence of three such pricings equal to three calls bit for bit, and the
    per-step flow's K-per-launch form (K = 4 at this size) equal to its single launches."""
    from options_model_amd import _ffi
    from oracle import cpu as orc
    M, N = 4_000_000, 252
    kw = dict(model="heston", is_put=False, n_steps=N, seed=42, heston_scheme="full_truncation", **HP)
    a = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=M, stream=9, **kw))
    b = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=M, stream=9, S0=200.0, K=200.0, **kw))
    assert b["price"] == 2.0 * a["price"] and b["n_exercised"] == a["n_exercised"] and b["sum_nitm"] == a["sum_nitm"]
    eur = ctx.price_european(_ffi.make_params(n_paths=M, stream=9, **kw))
    assert a["price"] >= eur["price"] - 3 * a["std"] / math.sqrt(M) and 9.5 < a["price"] < 11.5
    Mc = 200_000
    g = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=Mc, stream=9, **kw))
    So = orc.heston_paths(Mc, N, 100.0, 0.05, 1.0, HP["v0"], HP["kappa"], HP["theta"], HP["xi"], HP["rho"], 42, 9, 0, 1)
    ref = orc.lsm_poly(So, 100.0, 0.05, 1.0, False, "two_pass")
    assert g["price"] == pytest.approx(ref["price"], rel=1e-3)
    ps = [_ffi.make_params(semantics="two_pass", n_paths=M, stream=20 + i, **kw) for i in range(3)]
    for p, s in zip(ps, ctx.price_american_seq(ps)):
        one = ctx.price_american(p)
        assert (s["price"], s["sumsq"], s["n_exercised"], s["sum_nitm"]) == (one["price"], one["sumsq"], one["n_exercised"], one["sum_nitm"])
    ps = [_ffi.make_params(semantics="reference", n_paths=M, stream=30 + i, **kw) for i in range(4)]
    assert ctx.seq_step_width(ps) == 4
    for p, s in zip(ps, ctx.price_american_seq(ps)):
        one = ctx.price_american(p)
        assert (s["price"], s["sumsq"], s["n_exercised"], s["sum_nitm"]) == (one["price"], one["sumsq"], one["n_exercised"], one["sum_nitm"])


@pytest.mark.parametrize("model,M,sems", [("gbm", 1_000_000, ("two_pass", "reference", "textbook")),
                                          ("heston", 4_000_000, ("two_pass",))])
def test_full_size_pricing_is_exactly_homogeneous_in_spot_and_strike(ctx, model, M, sems):
    """Doubling S0 and K (a power of two: exact in binary floating point) doubles every float32 path value, leaves
    u = S/K - 1 and every regression moment of u unchanged and doubles every payoff sum -- so the WHOLE pricing
    (paths, moments, 3x3 solves, exercise decisions, valuation) must return exactly twice the price, the same
    exercise counts and the same regression-set sizes, bit for bit, at BASELINE's sizes (configs 2 and 4).  Any
    mis-indexed row, dropped path or order-dependent reduction breaks this."""
    from options_model_amd import _ffi
    for sem in sems:
        kw = dict(model=model, is_put=True, semantics=sem, n_paths=M, n_steps=252, seed=77, **(HP if model == "heston" else {}))
        a = ctx.price_american(_ffi.make_params(S0=100.0, K=100.0, **kw))
        b = ctx.price_american(_ffi.make_params(S0=200.0, K=200.0, **kw))
        c = ctx.price_american(_ffi.make_params(S0=25.0, K=25.0, **kw))
        for o, f in ((b, 2.0), (c, 0.25)):
            assert o["price"] == f * a["price"] and o["sum"] == f * a["sum"] and o["sumsq"] == f * f * a["sumsq"]
            assert (o["n_exercised"], o["n_zero"], o["sum_nitm"]) == (a["n_exercised"], a["n_zero"], a["sum_nitm"])
        assert a["n_exercised"] > M // 10


def test_config3_at_full_size_on_one_gpu(ctx):
    """BASELINE config 3 names 64M paths x 252 steps across 8 GPUs; the whole of it (65 GB of paths) also fits ONE
    MI355X.  Largest size the suite touches: 64-bit indexing everywhere, exact homogeneity (see above), agreement
    with the 1M-path pricing of the same contract within Monte-Carlo error, and the per-step flow's counters."""
    from options_model_amd import _ffi
    M, N = 64_000_000, 252
    a = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=5))
    b = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=5, S0=400.0, K=400.0))
    assert b["price"] == 4.0 * a["price"] and b["n_exercised"] == a["n_exercised"] and b["sum_nitm"] == a["sum_nitm"]
    small = ctx.price_american(_ffi.make_params(semantics="two_pass", n_paths=1_000_000, n_steps=N, seed=1234))
    se = math.sqrt(a["std"] ** 2 / M + small["std"] ** 2 / 1e6)
    assert abs(a["price"] - small["price"]) < 5 * se + 0.01
    assert a["n_paths"] == M and 0.4 * M < a["n_exercised"] < 0.7 * M and a["sum_nitm"] > 100 * M
    r = ctx.price_american(_ffi.make_params(semantics="reference", n_paths=M, n_steps=N, seed=5))
    assert 5.9 < r["price"] < 6.1 and r["n_exercised"] > 0.9 * M
    print(f"\nconfig 3 on one GPU: two-pass {a['ms_total']:.1f} ms = {M * N / a['ms_total'] / 1e-3:.3e} path-steps/s, "
          f"per-step flow {r['ms_total']:.1f} ms")


def test_pass1_time_partition_changes_no_number():
    """lsm_pass1_kernel picks the number of time steps per workgroup from the launch's size (one dispatch round when
    the launch fits the chip); partials are per (step, tile), so ANY partition must return the same bits.
    tools/soak_tchunk.py prices 120 problems (sizes around every threshold of that choice, GBM and Heston, puts and
    calls) under the automatic choice and two forced ones in separate processes and compares sums and counters."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_tchunk.py")], capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_the_maximum_number_of_time_steps(ctx):
    """n_steps = 4094, the most the library takes (4095 is refused): every flow once on a small path set against the
    oracles -- the fused polynomial pricings, "ols7" (whose pass 2 keeps its per-step table for the first 1,024 steps
    only: the rest takes the other branch), the rows of NN pass 1."""
    import torch

    from options_model_amd import _ffi
    from options_model_amd import nn_regressor as nnr
    from oracle import cpu as orc
    from oracle import reference_flow as rf
    from test_gpu_ols7 import _compare
    with pytest.raises(ValueError, match="exceeds the supported maximum"):
        ctx.price_american(_ffi.make_params(n_paths=64, n_steps=4095, seed=1))
    M, N, K, r, T = 512, 4094, 100.0, 0.05, 2.0
    for sem in ("two_pass", "reference", "textbook"):
        keep = ctx.empty((N + 1, M), np.float32)
        res = ctx.price_american(_ffi.make_params(semantics=sem, n_paths=M, n_steps=N, S0=100.0, K=K, r=r, sigma=0.3, T=T, seed=9), keep)
        Sg = keep.to_host()
        keep.free()
        So = orc.gbm_paths(M, N, 100.0, r, 0.3, T, 9, 0, 0, 1)
        assert np.abs(Sg / So - 1).max() <= 2e-4  # 4,094 float32 steps: the contract's 2e-5 is for 252
        ref = orc.lsm_poly(Sg, K, r, T, True, sem)
        assert res["price"] == pytest.approx(ref["price"], rel=1e-9)
        assert (res["n_exercised"], res["n_zero"], res["sum_nitm"]) == (ref["n_exercised"], ref["n_zero"], ref["sum_nitm"])
    S = ctx.gbm_paths(M, N, 100.0, r, 0.3, T, 9, 0)
    S32 = S.to_host()
    _compare(ctx, S, S32, K, r, T, True)
    S.free()
    St = torch.from_numpy(S32).cuda().contiguous()
    data, fm, fs, ym, ysd = nnr.build_rows_fused(St, K, r, T, True)
    itm = rf.payoff(S32[1:N].astype(np.float64), K, True) > 0
    assert data.shape[0] == int(itm.sum())
    # the rows of the LAST decision date come first (t descending), paths ascending: their first feature is x = S / K
    x_first = S32[N - 1][itm[N - 2]].astype(np.float64) / K
    got = data[: x_first.size, 1].double().cpu().numpy() * float(fs[1]) + float(fm[1])
    assert np.allclose(got, x_first, rtol=0, atol=5e-7)
    # NN pass 2 over all 4,093 decision dates, dropout on (the time step is part of the mask's Philox counter)
    torch.manual_seed(5)
    net = nnr.make_net(7, 64, 2, 0.1).cuda()
    rows = []
    disc = np.exp(-r * T / N)
    cfT = rf.payoff(S32[-1].astype(np.float64), K, True)
    for t in range(N - 1, 0, -1):
        cfT = cfT * disc
        sel = rf.payoff(S32[t].astype(np.float64), K, True) > 0
        if sel.any():
            rows.append((t, S32[t, sel].astype(np.float64), cfT[sel]))
    _, _, fm_o, fs_o, ym_o, ysd_o = rf.normalisers(rows, K, T, T / N)
    f64 = dict(dtype=torch.float64, device="cuda")
    hip = nnr.pass2_fused(St, K, r, T, True, net, torch.tensor(fm_o, **f64), torch.tensor(fs_o, **f64), torch.tensor(float(ym_o), **f64),
                          torch.tensor(float(ysd_o), **f64), dropout_on=True, want_state=True, seed=77)
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    regress, predict = rf.two_pass_frozen_mlp_regressor(K, T, N, state, fm_o, fs_o, float(ym_o), float(ysd_o),
                                                        dropout=dict(p=0.1, seed=77, hidden=64, layers=2))
    cf, ex, _ = rf.lsm_two_pass(S32.astype(np.float64), K, r, T, True, regress, predict)
    assert int(((hip["tex"] < N) != ex).sum()) <= 3
    sx = hip["sx"].astype(np.float64)
    cf_h = np.maximum(K - sx, 0) * np.exp(-r * (T / N) * (hip["tex"].astype(np.float64) - 1))
    assert int((np.abs(cf_h - cf) > 2e-5).sum()) <= 5
