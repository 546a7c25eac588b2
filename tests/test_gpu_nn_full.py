"""BASELINE config 5 at its own size -- GBM American put, 1M paths x 252 steps, SingleLSMNet(7, 64, 2) --
against the oracle (oracle/reference_flow.py: options_model_3.py:482-516 pass 1, :542-563 normalisers,
:615-651 pass 2) on the SAME paths.  Nothing here compares a 252-step price with the reference's 50-step runs.

What is compared, and on what:
  * full size (1M x 252): the row count R of the library's pass 1 with the oracle's in-the-money test on the
    very matrix the kernels read -- exact; the target mean over all 1.16e8 rows with a float64 numpy sum;
  * a 100k-path slice OF THE SAME PHILOX STREAM (pairs [0, 50k) of the 500k: columns [0, 50k) and
    [500k, 550k) of the full matrix): the library's rows + normalisers vs oracle.normalisers (R exact, means and
    standard deviations to the stated float64 tolerance); then the network the library trained on the FULL
    1.16e8 rows is frozen and the library's pass-2 kernel (float32 MFMA, dropout off) is compared decision by
    decision with oracle.two_pass_frozen_mlp_regressor through oracle.lsm_two_pass on that slice;
  * the same with the network as the reference runs it at inference -- dropout ACTIVE (options_model_3.py:637-640,
    SURVEY F5): mlp_apply_kernel with dropout on vs the oracle's sweep under the SAME masks (oracle/dropout.py keyed by
    the columns of the 1M-path matrix), decision by decision;
  * the full-size prices, eval mode and dropout on, against the oracle's slice prices within sampling error; the DROP
    that dropout at inference causes (7.54 -> 6.93 at this size) against the oracle's paired drop on the slice; the
    dropout-on price against the PyTorch sweep (nn_regressor.pass2, torch's own mask stream) on the same 1M paths
    within the noise of two independent mask draws.
"""
import numpy as np
import pytest

from oracle import reference_flow as rf

DROP_SEED = 2 ** 61 + 5  # the mask key of the dropout-on comparisons (pass2_fused draws one from torch otherwise)

pytestmark = pytest.mark.gpu

S0, K, R_, SIG, T = 100.0, 100.0, 0.05, 0.2, 1.0
M, N = 1_000_000, 252
SLICE_PAIRS = 50_000


@pytest.fixture(scope="module")
def c5():
    """One config-5 pricing by the library's kernels; keeps the path matrix, the trained net and the normalisers."""
    import torch

    from options_model_amd import nn_regressor as nr
    dev = torch.device("cuda", 0)
    ctx = nr._ctx_on_torch_stream(0)
    S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
    nr.generate_paths(ctx, S, dict(model="gbm"), S0, R_, SIG, T, 42, 0)
    out = nr.price_with_paths(S, K, R_, T, True, 43, nn_hidden=64, nn_layers=2, nn_dropout=0.1, nn_epochs=25,
                              nn_lr=1e-3, trainer="hip")
    P = M // 2
    cols = torch.cat([torch.arange(0, SLICE_PAIRS, device=dev), torch.arange(P, P + SLICE_PAIRS, device=dev)])
    S_slice = S[:, cols].contiguous()
    yield dict(torch=torch, nr=nr, S=S, S_slice=S_slice, out=out)
    del S, S_slice


def test_slice_is_the_same_philox_stream(c5):
    """Pairs [0, 50k) generated on their own (pair_offset 0, 100k paths) are bit for bit the columns taken from the
    1M-path matrix: the slice the oracle sees IS config 5's stream, not a look-alike."""
    torch, nr = c5["torch"], c5["nr"]
    S2 = torch.empty((N + 1, 2 * SLICE_PAIRS), dtype=torch.float32, device=c5["S"].device)
    nr.generate_paths(nr._ctx_on_torch_stream(0), S2, dict(model="gbm"), S0, R_, SIG, T, 42, 0)
    assert torch.equal(S2, c5["S_slice"])


def test_full_size_row_count_and_target_mean_match_the_oracle_rule(c5):
    """options_model_3.py:492 (`payoff > 0`, every path, every t in N-1..1) on the 1M x 252 matrix itself."""
    out = c5["out"]
    assert out["trainer"] == "hip" and out["pass2"] == "hip" and out["rows"] == "hip"
    S = c5["S"]
    R = 0
    ysum = 0.0
    dt = T / N
    payT = rf.payoff(S[N].cpu().numpy().astype(np.float64), K, True)
    for lo in range(1, N, 42):
        blk = S[lo:min(lo + 42, N)].cpu().numpy()
        itm = rf.payoff(blk.astype(np.float64), K, True) > 0
        R += int(itm.sum())
        for i in range(blk.shape[0]):  # :491 cash-flows discounted step by step = terminal payoff * disc^(N-t)
            ysum += float(payT[itm[i]].sum()) * np.exp(-R_ * dt * (N - (lo + i)))
    assert out["R"] == R and 1.0e8 < R < 1.3e8
    assert out["Y_mean"] == pytest.approx(ysum / R, rel=1e-9)  # float64 sums over 1.16e8 rows, different order


def _oracle_rows(S64):
    """Pass 1 of options_model_3.py:482-516 through the oracle's own sweep: capture what `regress` is handed."""
    got = {}

    def regress(rows):
        got["rows"] = rows
        return None

    rf.lsm_two_pass(S64, K, R_, T, True, regress, lambda m, t, s: None)
    return got["rows"]


def test_slice_rows_and_normalisers_match_oracle(c5):
    torch, nr = c5["torch"], c5["nr"]
    Ss = c5["S_slice"]
    data, fm, fs, ym, ysd = nr.build_rows_fused(Ss, K, R_, T, True)
    S64 = Ss.cpu().numpy().astype(np.float64)
    rows = _oracle_rows(S64)
    X_all, Y_all, ofm, ofs, oym, oys = rf.normalisers(rows, K, T, T / N)
    assert data.shape[0] == X_all.shape[0]  # R exact
    # float64 sums on both sides, different association (per-tile partials vs numpy pairwise): 1e-10 relative
    assert np.allclose(fm.cpu().numpy(), ofm, rtol=1e-10, atol=0)
    assert np.allclose(fs.cpu().numpy(), ofs, rtol=1e-10, atol=0)
    assert float(ym) == pytest.approx(oym, rel=1e-10) and float(ysd) == pytest.approx(oys, rel=1e-10)
    # the rows: the oracle's normalised features / targets in the oracle's order, cast to float32 (:570-571)
    ref = np.concatenate([(X_all - ofm) / ofs, (Y_all - oym) / oys], axis=1).astype(np.float32)
    got = data.cpu().numpy()
    assert np.abs(got - ref).max() <= 4e-6  # kernel: float32 features of a float32 spot; oracle: float64 then cast
    del X_all, Y_all, ref, got


def test_frozen_full_size_net_decisions_match_oracle_on_the_slice(c5):
    """The 2 x 64 network trained on all 1.16e8 rows, frozen: mlp_apply_kernel (dropout off) vs the oracle's
    pass 2 (options_model_3.py:615-651) on the 100k-path slice.  Exercise decisions may differ only where the
    immediate payoff sits within float32 rounding of the network's output (float32 MFMA sums vs numpy sgemm)."""
    torch, nr = c5["torch"], c5["nr"]
    out, Ss = c5["out"], c5["S_slice"]
    net, fm, fs = out["net"], out["feat_mean"], out["feat_std"]
    ym = torch.tensor(out["Y_mean"], dtype=torch.float64, device=Ss.device)
    ysd = torch.tensor(out["Y_std"], dtype=torch.float64, device=Ss.device)
    hip = nr.pass2_fused(Ss, K, R_, T, True, net, fm, fs, ym, ysd, dropout_on=False, want_state=True)
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    regress, predict = rf.two_pass_frozen_mlp_regressor(K, T, N, state, fm.cpu().numpy(), fs.cpu().numpy(),
                                                        out["Y_mean"], out["Y_std"])
    S64 = Ss.cpu().numpy().astype(np.float64)
    cf, ex, _ = rf.lsm_two_pass(S64, K, R_, T, True, regress, predict)
    ex_hip = hip["tex"] < N
    flips = int((ex_hip != ex).sum())
    # a path also "flips" silently when it exercises at another step: compare the realised cash-flows too
    dt = T / N
    pay = np.maximum(K - hip["sx"].astype(np.float64), 0)
    cf_hip = pay * np.exp(-R_ * dt * (hip["tex"].astype(np.float64) - 1))
    moved = int((np.abs(cf_hip - cf) > 2e-5).sum())
    price_o = float(cf.mean())
    print(f"config-5 slice: oracle price {price_o:.6f}  hip {hip['price']:.6f}  mask flips {flips}  "
          f"paths with another exercise time {moved} of {2 * SLICE_PAIRS}  exercised {ex.mean():.4f}")
    assert 0.3 < ex.mean() < 0.99  # real early exercise on the slice (measured 0.54)
    assert flips <= 3 and moved <= 10, (flips, moved)  # boundary paths only (measured on MI355X: 0 flips, 1 moved)
    assert abs(hip["price"] - price_o) <= 2e-5 * price_o, (hip["price"], price_o)  # measured 1.6e-6
    c5["slice_oracle"] = (price_o, float(cf.std()))
    c5["slice_cf_eval"] = cf


def test_dropout_on_decisions_match_oracle_on_the_slice_under_the_same_masks(c5):
    """The default mode (dropout left on at inference, F5).  The slice is priced as a SHARD of the 1M-path matrix
    (col_bases = (0, P): its columns keep the mask keys they have in the whole matrix), the oracle draws the same masks
    (oracle/dropout.py, keyed by the unsharded column and the time step) and runs the reference's sweep (:615-651) with
    the float32 numpy forward pass.  Same allowance as in eval mode."""
    torch, nr = c5["torch"], c5["nr"]
    out, Ss = c5["out"], c5["S_slice"]
    if "slice_cf_eval" not in c5:
        pytest.skip("needs the eval-mode slice comparison")
    net, fm, fs = out["net"], out["feat_mean"], out["feat_std"]
    P = M // 2
    ctx = nr._ctx_on_torch_stream(0)
    params = nr.flatten_params(net)
    torch.cuda.synchronize()
    hip = ctx.lsm_apply_mlp(Ss.data_ptr(), Ss.stride(0), 2 * SLICE_PAIRS, N, K, R_, T, True, params.data_ptr(),
                            fm.cpu().numpy(), fs.cpu().numpy(), out["Y_mean"], out["Y_std"], 0.1, DROP_SEED,
                            want_state=True, hidden=64, layers=2, col_bases=(0, P))
    cols_full = np.concatenate([np.arange(SLICE_PAIRS), np.arange(P, P + SLICE_PAIRS)])
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    regress, predict = rf.two_pass_frozen_mlp_regressor(
        K, T, N, state, fm.cpu().numpy(), fs.cpu().numpy(), out["Y_mean"], out["Y_std"],
        dropout=dict(p=0.1, seed=DROP_SEED, hidden=64, layers=2, col_of=lambda j: cols_full[j]))
    S64 = Ss.cpu().numpy().astype(np.float64)
    cf, ex, _ = rf.lsm_two_pass(S64, K, R_, T, True, regress, predict)
    ex_hip = hip["tex"] < N
    flips = int((ex_hip != ex).sum())
    dt = T / N
    pay = np.maximum(K - hip["sx"].astype(np.float64), 0)
    cf_hip = pay * np.exp(-R_ * dt * (hip["tex"].astype(np.float64) - 1))
    moved = int((np.abs(cf_hip - cf) > 2e-5).sum())
    price_o = float(cf.mean())
    d = cf - c5["slice_cf_eval"]  # the same 100k paths with and without the masks
    print(f"config-5 slice, dropout on: oracle price {price_o:.6f}  hip {hip['price']:.6f}  mask flips {flips}  moved {moved} "
          f"of {2 * SLICE_PAIRS}  exercised {ex.mean():.4f}; paired drop vs eval mode {d.mean():.4f} +- {d.std() / np.sqrt(d.size):.4f}")
    assert flips <= 3 and moved <= 10, (flips, moved)
    assert abs(hip["price"] - price_o) <= 2e-5 * price_o, (hip["price"], price_o)
    assert d.mean() < 0  # noise under the sticky rule only ever triggers EARLIER exercise: the oracle shows the drop too
    c5["slice_oracle_on"] = (price_o, float(cf.std()), float(d.mean()), float(d.std()))
    c5["slice_hip_on"] = hip


def test_full_size_prices_agree_with_the_oracle_slice_within_sampling_error(c5):
    torch, nr = c5["torch"], c5["nr"]
    out, S = c5["out"], c5["S"]
    if "slice_oracle" not in c5 or "slice_oracle_on" not in c5:
        pytest.skip("needs the slice comparisons of the previous tests")
    price_o, sd_o = c5["slice_oracle"]
    price_on_o, sd_on_o, drop_o, sd_drop_o = c5["slice_oracle_on"]
    ym = torch.tensor(out["Y_mean"], dtype=torch.float64, device=S.device)
    ysd = torch.tensor(out["Y_std"], dtype=torch.float64, device=S.device)
    args = (S, K, R_, T, True, out["net"], out["feat_mean"], out["feat_std"], ym, ysd)
    full_eval = nr.pass2_fused(*args, dropout_on=False)
    n_slice = 2 * SLICE_PAIRS
    se = sd_o / np.sqrt(n_slice)
    assert abs(full_eval["price"] - price_o) <= 4 * se, (full_eval["price"], price_o, se)
    # dropout on, the masks of DROP_SEED: the slice's columns inside the whole matrix decide exactly as the slice did
    # as a shard (the oracle-checked decisions are a sample of the full-size run, not a look-alike) ...
    full_on = nr.pass2_fused(*args, dropout_on=True, want_state=True, seed=DROP_SEED)
    P = M // 2
    cols_full = np.concatenate([np.arange(SLICE_PAIRS), np.arange(P, P + SLICE_PAIRS)])
    assert np.array_equal(full_on["tex"][cols_full], c5["slice_hip_on"]["tex"])
    assert np.array_equal(full_on["sx"][cols_full], c5["slice_hip_on"]["sx"])
    # ... the full-size dropout-on price agrees with the oracle's slice price within sampling error ...
    assert abs(full_on["price"] - price_on_o) <= 4 * sd_on_o / np.sqrt(n_slice), (full_on["price"], price_on_o)
    # ... and the DROP against eval mode (measured 7.54 -> 6.93) is the oracle's paired drop on the slice: a property of
    # the reference's F5 under its sticky rule with 252 decision dates, reproduced by the restatement -- not a bug
    drop_full = full_on["price"] - full_eval["price"]
    se_drop = sd_drop_o / np.sqrt(n_slice) * np.sqrt(1.0 + n_slice / M)
    print(f"config 5 full size: eval {full_eval['price']:.4f}  dropout on {full_on['price']:.4f}  drop {drop_full:.4f}; "
          f"oracle slice: eval {price_o:.4f}  dropout on {price_on_o:.4f}  paired drop {drop_o:.4f} +- {se_drop:.4f}")
    assert abs(drop_full - drop_o) <= 4 * se_drop, (drop_full, drop_o, se_drop)
    # the PyTorch sweep on the same 1M paths with torch's own mask stream: two independent mask draws on identical
    # paths differ by at most the sampling error of a difference of two such prices
    cf_t, _ = nr.pass2(*args, dropout_on=True)
    tol = 4 * np.sqrt(2.0) * float(cf_t.std()) / np.sqrt(M)
    assert abs(full_on["price"] - float(cf_t.mean())) <= tol, (full_on["price"], float(cf_t.mean()), tol)
    # the pricing's own result (mask key drawn from torch's generator) is one more draw of the same thing
    assert abs(out["price"] - full_on["price"]) <= tol, (out["price"], full_on["price"], tol)
    assert out["stderr"] < 0.02 and out["epochs_run"] >= 3


def test_config5_facade_is_reproducible(c5):
    """The drop-in call for config 5: every stage in the library's kernels, the same bits on a second call."""
    from options_model_amd import price_american_option
    res = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 1_000_000, 252, regressor="nn", seed=42)
    assert res.n_paths == 1_000_000 and res.info["trainer"] == "hip" and res.info["pass2"] == "hip"
    assert res.info["rows"] == "hip" and res.sum_nitm == c5["out"]["R"]
    again = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 1_000_000, 252, regressor="nn", seed=42)
    assert again.price == res.price
    assert abs(res.price - c5["out"]["price"]) <= 0.05 * res.price  # other torch seed for init / dropout
