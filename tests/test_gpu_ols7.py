"""Regressor "ols7": the v3 two-pass flow (options_model_3.py:482-516, 542-563, 615-651) with ONE least-squares fit on the
reference's seven features (:105-121) -- omc_lsm_ols7 (csrc/omc_ols7.hip: one co-moment sweep, a 6 x 6 solve on the host,
the sticky pass 2) against the oracle (oracle.reference_flow.two_pass_ols7_regressor: numpy lstsq on the materialised
design matrix) on the same float32 paths, against the fixture built with the reference's own feature function
(tests/golden/poly_flows.npz ols7_mid_put_*) and against SURVEY's anchor 6.986699434012439 on the reference's seed-42
paths."""
import numpy as np
import pytest

from oracle import cpu as orc
from oracle import reference_flow as rf

pytestmark = pytest.mark.gpu

K, R, SIG, T = 100.0, 0.05, 0.2, 1.0


def _oracle(S32, K_, r, T_, is_put):
    N = S32.shape[0] - 1
    reg, pred = rf.two_pass_ols7_regressor(K_, T_, N)
    return rf.lsm_two_pass(S32.astype(np.float64), K_, r, T_, is_put, reg, pred)


def _cashflows(out, K_, r, T_, N, is_put):
    sx = out["sx"].astype(np.float64)
    pay = np.maximum((K_ - sx) if is_put else (sx - K_), 0)
    return pay * np.exp(-r * (T_ / N) * (out["tex"].astype(np.float64) - 1))


def _compare(ctx, S_dev, S32, K_, r, T_, is_put, max_flips=2):
    N, M = S32.shape[0] - 1, S32.shape[1]
    out = ctx.lsm_ols7(S_dev, K_, r, T_, is_put, want_state=True)
    cf, ex, m = _oracle(S32, K_, r, T_, is_put)
    if m is None:  # never in the money: no regression, nobody exercises (:518-519)
        assert out["sum_nitm"] == 0 and np.all(out["weights"] == 0) and np.all(out["tex"] == N)
        assert out["price"] == pytest.approx(float(cf.mean()), rel=1e-12, abs=1e-300)
        return out, None
    assert out["sum_nitm"] == m["R"]
    # normalisers: float64 on both sides; the looser side is numpy's (a column mean over axis 0 of the R x 7 matrix is a
    # plain running sum: ~R eps), the kernel's merged triples are good to ~1e-15
    assert np.allclose(out["feat_mean"], m["fm"], rtol=1e-9, atol=0)
    # a constant column: the kernel's merged triples give variance exactly 0 (-> std 1, :562); numpy's np.std of the same
    # column may return rounding noise (4e-14 seen for s with one decision date) -- which the reference would divide by
    const = m["fs"] <= 1e-12 * np.maximum(np.abs(m["fm"]), 1e-300)
    assert np.all(out["feat_std"][const] == 1.0)
    assert np.allclose(out["feat_std"][~const], m["fs"][~const], rtol=1e-8, atol=0)
    assert out["y_mean"] == pytest.approx(float(m["Y_mean"]), rel=1e-9) and out["y_std"] == pytest.approx(float(m["Y_std"]), rel=1e-8)
    # the FIT is compared through what pass 2 uses, its predictions (the weights of the nearly collinear x, x^2, x^3 columns
    # move by 1e-7 between the normal equations and lstsq's SVD; predictions by 1e-10)
    ex_h = out["tex"] < N
    cf_h = _cashflows(out, K_, r, T_, N, is_put)
    assert out["price"] == pytest.approx(float(cf_h.mean()), rel=1e-9, abs=1e-300)  # the price is the mean of ITS decisions
    # How well is the fit determined at all?  x, x^2, x^3 over a narrow range of spots are nearly collinear: the standardised
    # design matrix of a deep out-of-the-money option (a few hundred rows, all within a few percent of the strike) has
    # singular values down to 1e-13 of the largest -- numpy's SVD and the kernel's 6 x 6 normal equations (which square the
    # condition number) then each return one of many near-solutions: in the case the fuzz soak found (profiles/r05_fuzz_soak.txt)
    # lstsq gave continuation values of 320 for payoffs of 7, the kernel 13.  Where cond > 1e4 only what is determined is
    # compared: the row count, the normalisers, the price as the mean of the kernel's own decisions.
    X = np.vstack([rf.regression_features(S32[t][rf.payoff(S32[t].astype(np.float64), K_, is_put) > 0].astype(np.float64), K_, T_,
                                          t * T_ / N) for t in range(N - 1, 0, -1)])
    # (a direction that is null BY CONSTRUCTION -- max(x - 1, 0) = x - 1 on every row of a call -- is harmless: both solvers
    # drop it and the predictions on in-the-money spots do not depend on how; relative singular value ~1e-17)
    rel = np.linalg.svd(((X - m["fm"]) / m["fs"])[:, ~const], compute_uv=False)
    rel = rel / rel[0] if rel.size and rel[0] > 0 else np.zeros(1)
    # (two cases of a 38,000-case soak had a smallest relative singular value of 6e-5 / 4e-6 and disagreed at the rows by
    # 2e-3 / 1e-5.  The kernel forms its co-moments in u = x - 1 since -- with that the second case agrees to 1e-7; in the
    # first -- 121 rows on one date, 1 on the other -- it is numpy's lstsq that moves by 2e-3 between the device's and the
    # oracle's float32 paths, while the kernel's value equals a long-double projection to 1e-8: such designs are compared
    # on what is determined only)
    if m["R"] < 100 or ((rel > 1e-13) & (rel < 1e-4)).any():
        return out, m
    # ... AT THE ROWS (a sample of them, every date represented): what the least-squares problem determines even when a
    # direction is null in this sample (a date with a single row: s and x s are then collinear) -- away from the rows two
    # exact solutions may differ by anything
    f = X[np.unique(np.linspace(0, X.shape[0] - 1, 256).astype(np.int64))]
    ref = ((f - m["fm"]) / m["fs"]) @ m["w"] * m["Y_std"] + m["Y_mean"]
    got = ((f - out["feat_mean"]) / out["feat_std"]) @ out["weights"] * out["y_std"] + out["y_mean"]
    assert np.allclose(got, ref, rtol=1e-7, atol=1e-7 * float(m["Y_std"]))
    flips = int((ex_h != ex).sum())
    moved = int((np.abs(cf_h - cf) > 2e-5).sum())
    assert flips <= max_flips and moved <= 2 * max_flips + 1, (flips, moved, M)
    # (a flipped path moves the price by its cash-flow difference / M: nothing to bound beyond the two counts)
    if not (flips or moved):
        assert out["price"] == pytest.approx(float(cf.mean()), rel=1e-9)
    return out, m


def test_fixture_built_with_the_references_feature_function(ctx, golden):
    """The reference's own paths (float64, rounded to the kernels' float32): R exact, the fit's predictions, the decisions and
    the price of the fixture."""
    g, pf = golden["paths"], golden["poly"]
    S64 = g["gbm_mid_S"]
    S32 = S64.astype(np.float32)
    S = ctx.to_device(S32, np.float32)
    out = ctx.lsm_ols7(S, K, R, T, True, want_state=True)
    S.free()
    N = S32.shape[0] - 1
    assert abs(out["sum_nitm"] - int(pf["ols7_mid_put_R"])) <= 2  # float32 rounding of a spot sitting on the strike
    assert np.allclose(out["feat_mean"], pf["ols7_mid_put_feat_mean"], rtol=2e-6, atol=1e-9)
    assert np.allclose(out["feat_std"], pf["ols7_mid_put_feat_std"], rtol=2e-5, atol=1e-9)
    flips = int(((out["tex"] < N) != pf["ols7_mid_put_ex"]).sum())
    assert flips <= 2, flips
    assert out["price"] == pytest.approx(float(pf["ols7_mid_put_cf"].mean()), rel=2e-3 if flips else 2e-5)


def test_anchor_on_the_references_seed42_paths(ctx, golden):
    """6.986699434012439 on the 10k x 50 paths the reference builds for RNGManager(42) (float64 there)."""
    z_half = rf.RNGManager(42).get_child_rng().standard_normal((50, 5000))
    S64 = rf.gbm_paths_from_normals(z_half, 100.0, R, SIG, T)
    S32 = S64.astype(np.float32)
    S = ctx.to_device(S32, np.float32)
    out, m = _compare(ctx, S, S32, K, R, T, True)
    S.free()
    assert out["price"] == pytest.approx(6.986699434012439, rel=2e-4)  # float32 paths against the float64 anchor
    assert abs(out["sum_nitm"] - 225_057) <= 2  # R of the float64 paths; float32 rounding of a spot sitting on the strike


@pytest.mark.parametrize("M,N,is_put,S0,sigma", [
    (20_000, 50, True, 100.0, 0.2), (20_000, 50, False, 100.0, 0.2), (4_098, 7, True, 100.0, 0.3), (130, 3, True, 100.0, 0.4),
    (2, 2, True, 100.0, 0.5), (100_000, 25, True, 90.0, 0.2), (50_000, 60, False, 110.0, 0.25), (10_000, 130, True, 100.0, 0.2),
])
def test_gbm_matches_the_oracle_on_the_same_paths(ctx, M, N, is_put, S0, sigma):
    S = ctx.gbm_paths(M, N, S0, R, sigma, T, 7, 3)
    _compare(ctx, S, S.to_host(), K, R, T, is_put)
    S.free()


def test_heston_matches_the_oracle_on_the_same_paths(ctx):
    S = ctx.heston_paths(40_000, 40, 100.0, R, T, 0.04, 2.0, 0.04, 0.3, -0.7, 11, 0, scheme=1)
    for is_put in (True, False):
        _compare(ctx, S, S.to_host(), K, R, T, is_put)
    S.free()


def test_never_in_the_money_and_constant_columns(ctx):
    # a put that is never in the money: no rows, no exercise, the discounted terminal payoff (0)
    S = ctx.gbm_paths(2_000, 10, 400.0, R, 0.1, T, 3, 0)
    out, m = _compare(ctx, S, S.to_host(), K, R, T, True)
    S.free()
    assert m is None and out["price"] == 0.0
    # N = 2: one decision date, the s column is constant (std 0 -> 1, weight 0) and x*s is collinear with x
    S = ctx.gbm_paths(5_000, 2, 100.0, R, 0.3, T, 5, 0)
    out, m = _compare(ctx, S, S.to_host(), K, R, T, True)
    S.free()
    assert out["feat_std"][5] == 1.0 and out["weights"][5] == 0.0 and out["weights"][0] == 0.0
    assert out["feat_std"][4] == 1.0 and out["weights"][4] == 0.0  # max(x - 1, 0) is identically 0 for a put


def test_facade_regressor_ols7(ctx):
    from options_model_amd import price_american_option
    r1 = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 50, regressor="ols7", seed=42, ctx=ctx)
    S = ctx.gbm_paths(200_000, 50, 100.0, 0.05, 0.2, 1.0, 42, 0)
    direct = ctx.lsm_ols7(S, 100.0, 0.05, 1.0, True)
    S.free()
    assert r1.price == direct["price"] and r1.info["regressor"] == "ols7" and len(r1.info["weights"]) == 7
    assert r1.semantics == "two_pass" and r1.sum_nitm == direct["sum_nitm"] and 0 < r1.stderr < 0.05
    poly = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 50, regressor="poly", seed=42, ctx=ctx)
    assert abs(r1.price - poly.price) < 0.6  # another regressor (one global fit, not one per step) on the same paths: same ballpark
    h = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 50_000, 40, model="Heston", option_type="call", regressor="ols7",
                              heston_scheme="full_truncation", seed=5, ctx=ctx)
    assert 8.0 < h.price < 13.0
    with pytest.raises(ValueError, match="'poly', 'nn' or 'ols7'"):
        price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 1000, 10, regressor="spline")


def test_collective_calls_fail_on_every_rank_together():
    """On a context with a communicator / hook omc_lsm_ols7 and the full omc_nn_build_rows are collective.  A failure only
    one rank can see travels as the ninth double of the first all-reduce: here the hook plays (a) a healthy peer with the
    same data (doubling: the job's fit is this rank's fit, the job has twice the rows) and (b) a peer that reports a
    failure -- this rank must come back with an error (3103 / 3102), not with a price, and only after it has taken part
    in that all-reduce.  And (c) a LOCAL failure (a row buffer too small for this rank's rows) is reported with the
    rank's own message after the all-reduce, not before it."""
    import torch

    from options_model_amd import _ffi
    from options_model_amd.dist import _DevPtr
    stream = torch.cuda.Stream()
    c = _ffi.Context(0, stream=stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            S = c.gbm_paths(20_000, 20, 100.0, R, SIG, T, 3, 0)
            alone = c.lsm_ols7(S, K, R, T, True)
            calls = []

            def doubling(dptr, count):
                calls.append(count)
                torch.as_tensor(_DevPtr(dptr, count), device="cuda").mul_(2.0)

            c.set_allreduce_hook(doubling)
            c.set_option("world_size", 2)
            both = c.lsm_ols7(S, K, R, T, True)
            assert calls == [9, 28, 8]
            assert both["sum_nitm"] == 2 * alone["sum_nitm"] and both["price"] == pytest.approx(alone["price"], rel=1e-12)
            assert np.allclose(both["weights"], alone["weights"], rtol=1e-9, atol=1e-12)
            calls.clear()

            def peer_failed(dptr, count):
                calls.append(count)
                t = torch.as_tensor(_DevPtr(dptr, count), device="cuda")
                if count == 9:
                    t[8] += 1.0  # the other rank's flag
                else:
                    t.mul_(2.0)

            c.set_allreduce_hook(peer_failed)
            with pytest.raises(_ffi.OmcError, match="another rank"):
                c.lsm_ols7(S, K, R, T, True)
            assert calls == [9]
            calls.clear()
            n = c.nn_build_rows(S.ptr, S.shape[1], 20_000, 20, K, R, T, True)  # the count alone is not collective
            assert calls == [] and n == alone["sum_nitm"]
            data = torch.empty((n, 8), dtype=torch.float32, device="cuda")
            with pytest.raises(_ffi.OmcError, match="another rank"):
                c.nn_build_rows(S.ptr, S.shape[1], 20_000, 20, K, R, T, True, data.data_ptr(), n)
            assert calls == [9]
            calls.clear()
            c.set_allreduce_hook(doubling)
            with pytest.raises(ValueError, match="row buffer smaller"):
                c.nn_build_rows(S.ptr, S.shape[1], 20_000, 20, K, R, T, True, data.data_ptr(), n - 1)
            assert calls == [9]  # it entered the all-reduce before it reported
            calls.clear()
            got = c.nn_build_rows(S.ptr, S.shape[1], 20_000, 20, K, R, T, True, data.data_ptr(), n)
            assert got[0] == n and calls == [9, 8]
            # (d) ADVICE r5 (medium): the fused call's OWN path matrix -- its largest allocation -- does not fit on this rank
            # alone (option "alloc_limit"): the rank must still enter the first all-reduce (its peers are in it) and
            # then report its own error; the same budget on a plain context fails at once, without any collective
            calls.clear()
            big = _ffi.make_params(semantics="two_pass", n_paths=400_000, n_steps=40, seed=3)  # 65.6 MB of paths
            c.set_option("alloc_limit", 32 << 20)
            with pytest.raises(_ffi.OmcError, match="alloc_limit"):
                c.price_american_ols7(big)
            assert calls == [9]
            c.set_option("alloc_limit", 0)
            calls.clear()
            ok = c.price_american_ols7(big)
            assert calls == [9, 28, 8] and 5.0 < ok["price"] < 9.0
            # (e) a peer whose PASS 2 failed: its flag arrives in slot 7 of the result sums (zero from the kernels)
            calls.clear()

            def peer_failed_late(dptr, count):
                calls.append(count)
                t = torch.as_tensor(_DevPtr(dptr, count), device="cuda")
                t.mul_(2.0)
                if count == 8:
                    t[7] += 1.0

            c.set_allreduce_hook(peer_failed_late)
            with pytest.raises(_ffi.OmcError, match="another rank of the job could not run its pass 2"):
                c.price_american_ols7(big)
            assert calls == [9, 28, 8]
            c.set_allreduce_hook(None)
            c.set_option("world_size", 1)
            c.set_option("alloc_limit", 32 << 20)
            bigger = _ffi.make_params(semantics="two_pass", n_paths=900_000, n_steps=40, seed=3)  # (the 65.6 MB matrix exists by now)
            with pytest.raises(_ffi.OmcError, match="alloc_limit"):
                c.price_american_ols7(bigger)  # one rank: no collective to enter, the error comes back directly
            c.set_option("alloc_limit", 0)
            S.free()
    finally:
        c.set_option("alloc_limit", 0)
        c.close()
