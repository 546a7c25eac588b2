"""SURVEY.md Appendix A, item by item, on the HIP kernels: the four-path case whose answers were worked out by hand from
the reference's loops (tests/helpers/quirk_cases.py: literals, produced by no code under test) put to the per-step
kernel, to pass 2 of the two-pass flow, to the row builder of pass 1 and to the path generators.
tests/test_quirks_cpu.py puts the same known answers to the oracles.  The item numbers are Appendix A's."""
import math

import numpy as np
import pytest

from helpers import quirk_cases as q

pytestmark = pytest.mark.gpu


def _cf(out, tval):
    pay = np.maximum(q.K - out["sx"].astype(np.float64), 0)
    return pay * np.exp(-q.R * q.T / q.N * (out["tex"] - tval))


def test_items_5_6_7_12_per_step_kernel_on_the_hand_built_case(ctx):
    """lsm_step_kernel (omc_lsm_apply_values, semantics "reference"): discount before the in-the-money test, strict '>'
    twice, the sticky mask, values at t = dt -- and v1's (mean, population std, P(cash-flow == 0)) return."""
    S, Cd = ctx.to_device(q.S.astype(np.float32)), ctx.to_device(q.CONT)
    out = ctx.lsm_apply_values(S, q.K, q.R, q.T, True, Cd, "reference")
    assert np.array_equal(out["tex"], q.PER_STEP_TEX)
    assert np.array_equal(out["sx"], np.array([95.0, 70.0, 95.0, 85.0], np.float32))
    assert np.allclose(_cf(out, 1), q.PER_STEP_CF, rtol=1e-15, atol=0)
    assert out["price"] == pytest.approx(q.PER_STEP_CF.mean(), rel=1e-14)
    assert out["std"] == pytest.approx(math.sqrt(((q.PER_STEP_CF - q.PER_STEP_CF.mean()) ** 2).sum() / 4), rel=1e-12)  # ddof = 0
    assert out["zero_prob"] == 0.0 and out["n_exercised"] == 3
    # (the set sizes -- in the money AND not yet exercised: 3, then 2 -- are what the tex above imply; values mode fits
    # nothing and reports no row count)
    # classic Longstaff-Schwartz on the same values: the other answer on every path
    tb = ctx.lsm_apply_values(S, q.K, q.R, q.T, True, Cd, "textbook")
    assert np.allclose(_cf(tb, 0), q.TEXTBOOK_CF, rtol=1e-15, atol=0)
    assert tb["price"] == pytest.approx(q.TEXTBOOK_CF.mean(), rel=1e-14)
    S.free(), Cd.free()


def test_items_6_7_pass2_of_the_two_pass_flow_on_the_hand_built_case(ctx):
    """lsm_pass2_kernel (omc_lsm_apply_frozen) with a continuation polynomial frozen to one constant per step
    (options_model_3.py:615-651): 15 > 15 does not exercise, the path on the strike is not in the money, an exercised
    path is never looked at again, values at t = dt."""
    S = ctx.to_device(q.S.astype(np.float32))
    b4 = np.zeros((q.N + 1, 4))
    b4[2] = [q.FROZEN_C[2], 0, 0, 3]
    b4[1] = [q.FROZEN_C[1], 0, 0, 3]
    out = ctx.lsm_apply_frozen(S, q.K, q.R, q.T, True, b4)
    S.free()
    assert np.array_equal(out["tex"], q.TWO_PASS_TEX)
    assert np.allclose(_cf(out, 1), q.TWO_PASS_CF, rtol=1e-15, atol=0)
    assert out["price"] == pytest.approx(q.TWO_PASS_CF.mean(), rel=1e-14)
    assert out["n_exercised"] == 3 and out["zero_prob"] == 0.0


def test_items_4_5_8_9_rows_of_pass1_on_the_hand_built_case(ctx):
    """omc_nn_build_rows (options_model_3.py:482-563): which (t, path) pairs become rows and in which order, the seven
    features of a row, the target = discounted TERMINAL payoff whatever happens in between, population std, zero std -> 1,
    and the float32 cast of the normalised matrix."""
    import torch
    from options_model_amd import nn_regressor as nr
    S = torch.from_numpy(q.S.astype(np.float32)).cuda().contiguous()
    data, fm, fs, ym, ysd = nr.build_rows_fused(S, q.K, q.R, q.T, True)
    assert data.shape == (6, 8) and data.dtype == torch.float32
    F = np.array([q.features(x, t * q.T / q.N) for t, x in zip(q.ROWS_T, q.ROWS_X)])
    fm, fs = fm.cpu().numpy(), fs.cpu().numpy()
    assert np.allclose(fm, F.mean(axis=0), rtol=1e-12, atol=1e-15)
    sd = np.sqrt(((F - F.mean(axis=0)) ** 2).sum(axis=0) / 6)           # / n
    assert fs[0] == 1.0 and fs[4] == 1.0                                # the constant column and max(x - 1, 0) of a put
    live = [1, 2, 3, 5, 6]
    assert np.allclose(fs[live], sd[live], rtol=1e-10, atol=0)
    assert float(ym) == pytest.approx(q.ROWS_Y.mean(), rel=1e-13)
    assert float(ysd) == pytest.approx(math.sqrt(((q.ROWS_Y - q.ROWS_Y.mean()) ** 2).sum() / 6), rel=1e-12)
    d = data.cpu().numpy().astype(np.float64)
    assert np.allclose(d[:, :7] * fs + fm, F, rtol=0, atol=4e-7)        # float32 normalised features, row by row in order
    assert np.allclose(d[:, 7] * float(ysd) + float(ym), q.ROWS_Y, rtol=0, atol=4e-6)
    assert np.all(d[:, 0] == 0.0) and np.all(d[:, 4] == 0.0)
    # the polynomial pass 1 sees the same sets: three in-the-money rows at each date
    Sd = ctx.to_device(q.S.astype(np.float32))
    out = ctx.lsm_poly(Sd, q.K, q.R, q.T, True, "two_pass")
    Sd.free()
    assert list(out["nitm"][1:3]) == [3, 3] and out["sum_nitm"] == 6


def test_items_1_2_even_path_count_and_the_antithetic_partner(ctx):
    z = np.array([[0.3, -1.2], [0.0, 2.0], [-0.7, 0.1]], np.float32)
    Sd = ctx.gbm_paths_from_normals(z, 100.0, 0.05, 0.2, 1.0)
    S = Sd.to_host().astype(np.float64)
    Sd.free()
    assert S.shape == (4, 4) and np.all(S[0] == 100.0)
    dt = 1.0 / 3
    drift, vol = (0.05 - 0.5 * 0.2 ** 2) * dt, 0.2 * math.sqrt(dt)
    zz = z.astype(np.float64)
    for j in range(2):
        up = 100.0 * np.exp(np.cumsum(drift + vol * zz[:, j]))
        dn = 100.0 * np.exp(np.cumsum(drift - vol * zz[:, j]))       # column j + M / 2: the same normals, negated
        assert np.allclose(S[1:, j], up, rtol=3e-7) and np.allclose(S[1:, j + 2], dn, rtol=3e-7)
    from options_model_amd import price_american_option
    a = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 2_001, 10, seed=3, ctx=ctx)  # options_model_3.py:458
    assert a.n_paths == 2_000


def test_item_3_heston_correlation_clamps_and_previous_variance(ctx):
    z1, z2 = np.array([[-3.0], [0.5]], np.float32), np.array([[-2.0], [1.0]], np.float32)
    kw = dict(v0=0.01, kappa=1.0, theta=0.02, xi=1.5, rho=-0.6)
    Sd = ctx.heston_paths_from_normals(z1, z2, 100.0, 0.03, 1.0, scheme=0, **kw)
    S = Sd.to_host().astype(np.float64)
    Sd.free()
    dt = 0.5
    for c, sgn in enumerate((1.0, -1.0)):
        v, s = kw["v0"], 100.0
        for t in range(2):
            a, b = sgn * float(z1[t, 0]), sgn * float(z2[t, 0])
            w2 = kw["rho"] * a + math.sqrt(1 - kw["rho"] ** 2) * b
            vp = max(v, 0.0)                                           # clamp before use ...
            v = max(v + kw["kappa"] * (kw["theta"] - vp) * dt + kw["xi"] * math.sqrt(vp * dt) * w2, 0.0)  # ... and at store
            s = s * math.exp((0.03 - 0.5 * vp) * dt + math.sqrt(vp * dt) * a)   # the spot moves with the PREVIOUS variance
            assert S[t + 1, c] == pytest.approx(s, rel=1e-6)
    # the antithetic path's variance is clamped to 0 after its first step: its second step is exp(r dt), no diffusion
    assert S[2, 1] / S[1, 1] == pytest.approx(math.exp(0.03 * dt), rel=5e-7)
