"""GPU tests of the NN continuation-value regressor (BASELINE config 5)
against the fixtures captured from the reference's own run (tests/golden/v3_frozen_nn.npz:
its trained weights, normalisers and eval-mode pass-2 decisions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _load_net(torch, nn, tag, hidden):
    from options_model_amd import nn_regressor as nr
    net = nr.make_net(7, int(hidden), 3, 0.1)
    state = {k[len(tag) + 4:]: torch.from_numpy(nn[k]) for k in nn.files if k.startswith(f"{tag}_sd_")}
    net.load_state_dict(state)
    return net.cuda()


@pytest.mark.parametrize("tag", ["gbm_put", "heston_call"])
def test_pass1_rows_and_normalisers_match_reference(torch_cuda, golden, tag):
    torch = torch_cuda
    from options_model_amd import nn_regressor as nr
    nn = golden["nn"]
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    S = torch.from_numpy(nn[f"{tag}_S"]).cuda()  # float64: the reference's own paths
    N = S.shape[0] - 1
    x, t, y, _ = nr.collect_rows(S, K, r, T, bool(is_put))
    assert x.numel() == int(nn[f"{tag}_R"])
    fm, fs, ym, ysd = nr.normalisers(x, t, y, T, T / N)
    assert np.allclose(fm.cpu().numpy(), nn[f"{tag}_feat_mean"], rtol=1e-12, atol=1e-14)
    assert np.allclose(fs.cpu().numpy(), nn[f"{tag}_feat_std"], rtol=1e-10, atol=1e-14)
    assert float(ym) == pytest.approx(nn[f"{tag}_Y_mean_std"][0], rel=1e-12)
    assert float(ysd) == pytest.approx(nn[f"{tag}_Y_mean_std"][1], rel=1e-12)


@pytest.mark.parametrize("tag", ["gbm_put", "heston_call"])
def test_hip_rows_and_normalisers_match_reference(torch_cuda, golden, tag):
    """omc_nn_build_rows (count, scan, statistics and row kernels) on the reference's paths rounded to
    float32: the reference's row count R and its normalisers; the rows themselves equal the ones the
    PyTorch restatement builds from the same float32 matrix, in the same order."""
    torch = torch_cuda
    from options_model_amd import nn_regressor as nr
    nn = golden["nn"]
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    S = torch.from_numpy(nn[f"{tag}_S"]).float().cuda().contiguous()
    N = S.shape[0] - 1
    data, fm, fs, ym, ysd = nr.build_rows_fused(S, K, r, T, bool(is_put))
    assert abs(data.shape[0] - int(nn[f"{tag}_R"])) <= 2  # float32 rounding of a spot sitting on the strike
    assert np.allclose(fm.cpu().numpy(), nn[f"{tag}_feat_mean"], rtol=2e-6, atol=1e-9)
    assert np.allclose(fs.cpu().numpy(), nn[f"{tag}_feat_std"], rtol=2e-5, atol=1e-9)
    assert float(ym) == pytest.approx(nn[f"{tag}_Y_mean_std"][0], rel=2e-6)
    assert float(ysd) == pytest.approx(nn[f"{tag}_Y_mean_std"][1], rel=2e-6)
    x, t, y, _ = nr.collect_rows(S, K, r, T, bool(is_put))
    fm2, fs2, ym2, ysd2 = nr.normalisers(x, t, y, T, T / N)
    ref = nr.build_training_matrix(x, t, y, fm2, fs2, ym2, ysd2, T, T / N)
    assert ref.shape == data.shape
    assert np.allclose(fm.cpu().numpy(), fm2.cpu().numpy(), rtol=1e-12, atol=1e-15)
    assert np.allclose(fs.cpu().numpy(), fs2.cpu().numpy(), rtol=1e-9, atol=1e-15)
    assert float((data - ref).abs().max()) <= 2e-6  # same order, float32 rounding of the last operation


def test_hip_rows_ragged_sizes_and_empty(torch_cuda):
    torch = torch_cuda
    from options_model_amd import nn_regressor as nr
    dev = torch.device("cuda", 0)
    c2 = nr._ctx_on_torch_stream(0)
    for M, N in ((2, 2), (254, 3), (258, 5), (1000, 33), (4096, 7)):
        S = torch.empty((N + 1, M), dtype=torch.float32, device=dev)
        nr.generate_paths(c2, S, dict(model="gbm"), 100.0, 0.05, 0.3, 1.0, 5)
        built = nr.build_rows_fused(S, 100.0, 0.05, 1.0, True)
        x, t, y, _ = nr.collect_rows(S, 100.0, 0.05, 1.0, True)
        assert built is not None and built[0].shape[0] == x.numel()
        fm2, fs2, ym2, ysd2 = nr.normalisers(x, t, y, 1.0, 1.0 / N)
        ref = nr.build_training_matrix(x, t, y, fm2, fs2, ym2, ysd2, 1.0, 1.0 / N)
        assert float((built[0] - ref).abs().max()) <= 5e-6, (M, N)
    S = torch.full((4, 64), 200.0, dtype=torch.float32, device=dev)  # a put that is never in the money
    assert nr.build_rows_fused(S, 100.0, 0.05, 1.0, True) is None


@pytest.mark.parametrize("tag", ["gbm_put", "heston_call"])
def test_pass2_with_reference_weights_reproduces_reference_decisions(torch_cuda, golden, tag):
    torch = torch_cuda
    from options_model_amd import nn_regressor as nr
    nn = golden["nn"]
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    S = torch.from_numpy(nn[f"{tag}_S"]).cuda()
    net = _load_net(torch, nn, tag, hidden)
    fm = torch.from_numpy(nn[f"{tag}_feat_mean"]).cuda()
    fs = torch.from_numpy(nn[f"{tag}_feat_std"]).cuda()
    ym, ysd = (torch.tensor(v, dtype=torch.float64, device="cuda") for v in nn[f"{tag}_Y_mean_std"])
    cf, ex = nr.pass2(S, K, r, T, bool(is_put), net, fm, fs, ym, ysd, dropout_on=False)
    flips = int((ex.cpu().numpy() != nn[f"{tag}_ex_eval"]).sum())
    assert flips <= 3, flips  # GPU GEMM summation order vs CPU torch: boundary paths only
    ref = float(nn[f"{tag}_price_eval"])
    assert abs(float(cf.mean()) - ref) <= 2e-3 * ref
    same = ex.cpu().numpy() == nn[f"{tag}_ex_eval"]
    assert np.allclose(cf.cpu().numpy()[same], nn[f"{tag}_cf_eval"][same], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tag", ["gbm_put", "heston_call"])
def test_hip_pass2_with_reference_weights_reproduces_reference_decisions(torch_cuda, golden, tag):
    """The library's own pass-2 kernel (omc_lsm_apply_mlp, float32 MFMA) on the reference's trained
    network (3 x 128 for gbm_put -- its default shape -- and 3 x 64 for heston_call), normalisers and
    paths: the reference's eval-mode exercise decisions and price come back.  Paths go through
    float32 here (the kernels' storage type), so a path whose payoff sits within rounding of its
    continuation value may flip."""
    torch = torch_cuda
    from options_model_amd import nn_regressor as nr
    nn = golden["nn"]
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    S = torch.from_numpy(nn[f"{tag}_S"]).float().cuda().contiguous()
    net = _load_net(torch, nn, tag, hidden)
    assert nr.fused_apply_supports(net) and nr.fused_trainer_supports(net, 256)
    fm = torch.from_numpy(nn[f"{tag}_feat_mean"]).cuda()
    fs = torch.from_numpy(nn[f"{tag}_feat_std"]).cuda()
    ym, ysd = (torch.tensor(v, dtype=torch.float64, device="cuda") for v in nn[f"{tag}_Y_mean_std"])
    out = nr.pass2_fused(S, K, r, T, bool(is_put), net, fm, fs, ym, ysd, dropout_on=False, want_state=True)
    N = S.shape[0] - 1
    ex = out["tex"] < N
    flips = int((ex != nn[f"{tag}_ex_eval"]).sum())
    assert flips <= 3, flips
    ref = float(nn[f"{tag}_price_eval"])
    assert abs(out["price"] - ref) <= 2e-3 * ref
    same = ex == nn[f"{tag}_ex_eval"]
    disc = np.exp(-r * (T / N) * (out["tex"].astype(np.float64) - 1))
    pay = np.maximum((K - out["sx"].astype(np.float64)) if is_put else (out["sx"].astype(np.float64) - K), 0)
    assert np.allclose((pay * disc)[same], nn[f"{tag}_cf_eval"][same], rtol=0, atol=2e-5)  # float32 S: half an ulp at S ~ 150 is 8e-6
    # the continuation values themselves: replay the reference's recorded eval-mode outputs
    on = nr.pass2_fused(S, K, r, T, bool(is_put), net, fm, fs, ym, ysd, dropout_on=True)
    assert abs(on["price"] - float(nn[f"{tag}_price_ref"])) <= 0.05 * ref  # other dropout stream


def test_hip_pass2_heston_put_reproduces_reference_decisions(torch_cuda, golden):
    """The Heston PUT run of the reference's v3 pricer (tests/golden/v3_frozen_nn_heston_put.npz): with its
    trained 3 x 64 network, normalisers and paths, the library's pass-2 kernel returns its eval-mode
    exercise decisions -- real early exercise under stochastic volatility (43 % of paths), which the
    Heston call fixture cannot show."""
    torch = torch_cuda
    from options_model_amd import nn_regressor as nr
    nn = golden["nn_heston_put"]
    tag = "heston_put"
    S0, K, r, sig, T, is_put, hidden = nn[f"{tag}_params"]
    assert bool(is_put)
    S = torch.from_numpy(nn[f"{tag}_S"]).float().cuda().contiguous()
    net = _load_net(torch, nn, tag, hidden)
    fm = torch.from_numpy(nn[f"{tag}_feat_mean"]).cuda()
    fs = torch.from_numpy(nn[f"{tag}_feat_std"]).cuda()
    ym, ysd = (torch.tensor(v, dtype=torch.float64, device="cuda") for v in nn[f"{tag}_Y_mean_std"])
    out = nr.pass2_fused(S, K, r, T, True, net, fm, fs, ym, ysd, dropout_on=False, want_state=True)
    N = S.shape[0] - 1
    ex = out["tex"] < N
    assert 0.2 < nn[f"{tag}_ex_eval"].mean() < 0.99
    assert int((ex != nn[f"{tag}_ex_eval"]).sum()) <= 3
    ref = float(nn[f"{tag}_price_eval"])
    assert abs(out["price"] - ref) <= 2e-3 * ref


def test_config1_nn_end_to_end_band(torch_cuda, golden):
    """10k x 50 ATM put, reference hyper-parameters (128x3, 25 epochs, batch 256).  The
    reference's own answer moves from seed to seed (6.81 for RNGManager(42); 7.29 / 6.96 / 7.19
    ... for seeds 1, 2, 3: tools/capture_reference_band.py) because the trained net, the
    dropout left on at inference (F5) and the look-ahead rule (F2) all feed the price.  Ours
    uses Philox paths and torch's GPU generator, so parity is membership in that band."""
    from options_model_amd import AdvancedOptionPricer, RNGManager
    sc = golden["scalars"]
    refs = [sc["end_to_end_10k_x_50_seed42"]["gbm_put_cv_off"]] + list(sc["reference_nn_seed_band"].values())
    assert len(refs) >= 4
    sd = float(np.std(refs, ddof=1))  # the reference's own seed-to-seed standard deviation: the band a further
    lo, hi = min(refs) - sd, max(refs) + sd  # reference seed would be held to (no fixed widening)
    p = AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(42),
                             use_control_variate=False)  # reference-only arguments: regressor defaults to "nn"
    price = p.price_american_option(100.0, 1.0, 10000, 50)
    info = p.last_result
    assert info["R"] > 200_000 and info["batch"] == 256 and info["n_paths"] == 10000
    assert info["trainer"] == "hip" and info["trainer_kernel"] == "mlp_train_q16_kernel" and info["pass2"] == "hip"
    assert lo < price < hi, (price, refs)
    assert p.rng_manager.get_child_seed() == golden["scalars"]["rng_manager_42_child_seeds"][2]


def test_config1_nn_prices_follow_the_references_distribution(torch_cuda, golden):
    """The price of ONE seed is a chaotic function of the training noise: on the same paths, rows and initial weights
    PyTorch autograd, the 32-row and the 16-row trainer kernels end at the same loss (to 1e-3) and at prices 7.02 / 7.14 /
    6.87 for seed 42, 6.96 / 6.69 / 6.45 for seed 5, 6.61 / 6.96 / 6.98 for seed 6 (profiles/r05_nn_seed_study.jsonl) --
    two equally good fits place the exercise boundary differently.  So prices cannot be compared seed by seed, and a band
    drawn from five reference runs (rounds 3-4) says little about a sixth.  Round 5: the reference itself was run for
    master seeds 42 and 1 .. 20 (tools/capture_reference_band.py, ~150 s of CPU each; tests/golden/scalars.json), and
    OUR pricer is run for the same 21 seeds.  The two SAMPLES must agree:
      * means: Welch's t below 3.0 (measured: -1.4 over the 21 seeds; 0.4 for the 3 x 64 net);
      * distributions: two-sample Kolmogorov-Smirnov statistic below its 0.1 % critical value 1.95 sqrt((n + m) / (n m));
      * spread: standard deviations within a factor 2 of each other;
      * every single price inside the reference's own range widened by its standard deviation -- the band a further
        reference seed would be held to."""
    from options_model_amd import AdvancedOptionPricer, RNGManager
    sc = golden["scalars"]
    ref = {42: sc["end_to_end_10k_x_50_seed42"]["gbm_put_cv_off"]}
    ref.update({int(k): v for k, v in sc["reference_nn_seed_band"].items()})
    assert len(ref) >= 15, "run tools/capture_reference_band.py for more seeds"
    seeds = sorted(ref)
    refs = np.array([ref[k] for k in seeds])
    prices = []
    for seed in seeds:
        p = AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(seed),
                                 use_control_variate=False)
        prices.append(p.price_american_option(100.0, 1.0, 10000, 50))
    _same_distribution(np.array(prices), refs, "config-1 NN 3 x 128")


def _same_distribution(prices, refs, what):
    """Two samples of prices over the same master seeds (ours / the reference's own runs): Welch's t below 3.0, the
    two-sample Kolmogorov-Smirnov statistic below its 0.1 % critical value, standard deviations within a factor 2, and
    every single price of ours inside the reference's range widened by its own standard deviation."""
    from scipy import stats
    n, m = len(prices), len(refs)
    t = (prices.mean() - refs.mean()) / np.sqrt(prices.var(ddof=1) / n + refs.var(ddof=1) / m)
    ks = stats.ks_2samp(prices, refs).statistic
    print(f"{what}, {n} seeds: ours mean {prices.mean():.4f} sd {prices.std(ddof=1):.4f} [{prices.min():.3f}, {prices.max():.3f}]; "
          f"reference mean {refs.mean():.4f} sd {refs.std(ddof=1):.4f} [{refs.min():.3f}, {refs.max():.3f}]; Welch t {t:.2f}, KS {ks:.3f}")
    assert abs(t) < 3.0, (t, prices, refs)
    assert ks < 1.95 * np.sqrt((n + m) / (n * m)), (ks, prices, refs)
    assert 0.5 < prices.std(ddof=1) / refs.std(ddof=1) < 2.0
    sd = refs.std(ddof=1)
    assert np.all((prices > refs.min() - sd) & (prices < refs.max() + sd)), (prices, refs)


def test_config1_nn_kernels_against_autograd_paired_over_seeds(torch_cuda):
    """the two-sample test above compares our prices with the reference's over 21 seeds at a price sd of 0.2 -- a
    bias of 2 % in the 16-row trainer could hide in it.  The PAIRED form removes the path noise: for each of 16 seeds the
    same paths, the same rows and the same initial weights (torch.manual_seed before the net is built), trained once by the
    library's kernels and once by PyTorch autograd + torch.optim.Adam (the reference's own training loop).  Minibatch
    order and dropout draws differ between the two (their generators cannot be matched), so single prices differ by the
    +-0.3 of profiles/r05_nn_seed_study.jsonl; what must hold:
      * every seed ends at the same loss: best epoch-mean loss within 1 %;
      * no systematic offset: |mean of the price differences| below 3 standard errors of that mean (paired t)."""
    from options_model_amd import nn_regressor as nnr
    d, rel_loss = [], []
    for seed in range(101, 117):
        kw = dict(seed=seed, torch_seed=seed + 1000, nn_hidden=128, nn_layers=3, nn_epochs=25)
        a = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 10_000, 50, nn_trainer="hip", **kw)
        b = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 10_000, 50, nn_trainer="torch", **kw)
        assert a.info["trainer"] == "hip" and a.info["trainer_kernel"] == "mlp_train_q16_kernel" and b.info["trainer"] == "torch"
        assert a.sum_nitm == b.sum_nitm and a.info["batch"] == b.info["batch"] == 256  # the same rows
        d.append(a.price - b.price)
        rel_loss.append(a.info["best_loss"] / b.info["best_loss"] - 1.0)
    d, rel_loss = np.array(d), np.array(rel_loss)
    t = d.mean() / (d.std(ddof=1) / np.sqrt(len(d)))
    print(f"paired over {len(d)} seeds: kernels - autograd: mean {d.mean():+.4f}, sd {d.std(ddof=1):.4f}, paired t {t:+.2f}; "
          f"best loss kernels / autograd - 1: mean {rel_loss.mean():+.2e}, max |.| {np.abs(rel_loss).max():.2e}")
    assert np.abs(rel_loss).max() < 1e-2, rel_loss
    assert abs(t) < 3.0, (t, d)


def test_config1_nn_hidden64_all_hip_follows_the_references_distribution(torch_cuda, golden):
    """Same flow with nn_hidden=64: SingleLSMNet(7, 64, 3) -- trained by the library's 16-row-tile MFMA trainer and
    applied by its pass-2 kernel (no PyTorch autograd, no PyTorch forward).  The reference's own prices for this setting
    (tools/capture_reference_band.py --hidden 64, ~75 s of CPU each; master seeds 1 .. 20 in
    scalars.json["reference_nn_seed_band_h64"]) against ours for the same master seeds, as two samples -- see
    test_config1_nn_prices_follow_the_references_distribution for why not seed by seed."""
    from options_model_amd import AdvancedOptionPricer, RNGManager
    band = golden["scalars"]["reference_nn_seed_band_h64"]
    seeds = sorted(int(k) for k in band)
    assert len(seeds) >= 4
    refs = np.array([band[str(k)] for k in seeds])
    prices = []
    for seed in seeds:
        p = AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(seed),
                                 use_control_variate=False, regressor="nn", nn_hidden=64)
        prices.append(p.price_american_option(100.0, 1.0, 10000, 50))
        info = p.last_result
        assert info["trainer"] == "hip" and info["pass2"] == "hip" and info["batch"] == 256
    _same_distribution(np.array(prices), refs, "config-1 NN 3 x 64")


def test_facade_nn_regressor_2x64(torch_cuda):
    from options_model_amd import price_american_option
    res = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 40_000, 25, regressor="nn", seed=3)
    assert 5.6 < res.price < 8.0 and res.n_paths == 40_000 and res.sum_nitm > 0


def test_a_shape_outside_the_kernels_says_so_once(torch_cuda):
    """`trainer="auto"` with a width the HIP kernels do not cover (96 units) trains and sweeps through PyTorch-ROCm: correct,
    slower -- and no longer silent: one RuntimeWarning per process and kind, info["trainer"] / ["pass2"] every time."""
    import warnings
    from options_model_amd import nn_regressor as nnr
    nnr._warned.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        outs = [nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 10, seed=3 + i, nn_hidden=96, nn_epochs=2)
                for i in range(2)]
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning) and "options_model_amd" in str(x.message)]
    assert len(msgs) == 2 and sum("HIP trainer kernels" in m for m in msgs) == 1 and sum("pass-2" in m for m in msgs) == 1
    assert all(o.info["trainer"] == "torch" and o.info["pass2"] == "torch" and 4.0 < o.price < 9.0 for o in outs)
    with warnings.catch_warnings(record=True) as w:  # a covered shape stays quiet
        warnings.simplefilter("always")
        o = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 10, seed=3, nn_hidden=64, nn_epochs=2)
    assert o.info["trainer"] == "hip" and not [x for x in w if "options_model_amd" in str(x.message)]


def test_nn_hidden_32_runs_in_the_kernels(torch_cuda):
    """Round 6: SingleLSMNet(7, 32, 3) -- a width the reference's constructor accepts (options_model_3.py:340-358,
    nn_hidden) -- trains through mlp_train_quad_kernel<32, 3> and sweeps through mlp_apply_kernel<32, 3>: no PyTorch
    autograd, no warning; same optimisation problem as the autograd trainer (same rows, same initial weights)."""
    import warnings
    from options_model_amd import AdvancedOptionPricer, RNGManager
    from options_model_amd import nn_regressor as nnr
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        p = AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(42),
                                 use_control_variate=False, nn_hidden=32)
        price = p.price_american_option(100.0, 1.0, 10000, 50)
    info = p.last_result
    assert info["trainer"] == "hip" and info["trainer_kernel"] == "mlp_train_quad_kernel" and info["pass2"] == "hip"
    assert 6.3 < price < 7.6 and not [x for x in w if "options_model_amd" in str(x.message)]
    kw = dict(seed=9, torch_seed=10, nn_hidden=32, nn_layers=3, nn_epochs=8, inference_dropout=False)
    a = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, nn_trainer="hip", **kw)
    b = nnr.price_american_option_nn(100.0, 100.0, 0.05, 0.2, 1.0, 20_000, 25, nn_trainer="torch", **kw)
    assert a.info["best_loss"] == pytest.approx(b.info["best_loss"], rel=1e-2) and abs(a.price - b.price) < 0.8


def test_bench_c1nn_line_is_the_references_default_call(torch_cuda):
    """`python bench.py --config c1nn`: ONE JSON line for the reference's default call (3 x 128, minibatch 256, dropout on)
    with the contract's fields, the trainer's MFMA roofline and the reference's own time as the quoted CPU baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c1nn", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["unit"] == "path-steps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(10000 * 50 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert d["info"]["trainer"] == "hip" and d["info"]["trainer_kernel"] == "mlp_train_q16_kernel" and d["info"]["batch"] == 256
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and 0 < r["frac"] < 0.1 and 20_000 < r["optimizer_steps"] <= 22_100
    assert 10 < r["us_per_optimizer_step"] < 60
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["value"] == pytest.approx(10000 * 50 / 133.0) and c["cores"] == 8
    assert 6.0 < d["price"] < 8.0 and d["ms_per_step"] < 2000
