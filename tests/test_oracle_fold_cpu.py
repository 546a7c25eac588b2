"""The folded-storage oracle (oracle/omc_oracle.c: orc_lsm_two_pass_folded) pinned on the CPU: against a plain numpy
two-pass flow over an explicit float64 matrix [S, C / S], and against the full-matrix oracle (itself pinned by
tests/golden/poly_flows.npz) on the float32 antithetic matrix the folded storage stands for."""
import numpy as np
import pytest

from oracle import cpu as orc


def numpy_two_pass(S64, K, r, T, is_put):
    """options_model_3.py:482-516 + :615-651 with OLS on [1, u, u^2]; S64 [N+1][M] float64"""
    N, M = S64.shape[0] - 1, S64.shape[1]
    dt = T / N
    pay = (K - S64) if is_put else (S64 - K)
    u = S64 / K - 1.0
    yN = np.maximum(pay[N], 0.0)
    betas, nitm = np.zeros((N + 1, 3)), np.zeros(N + 1, np.int64)
    for t in range(1, N):
        itm = pay[t] > 0
        nitm[t] = itm.sum()
        if nitm[t] >= 3:
            A = np.stack([np.ones(nitm[t]), u[t, itm], u[t, itm] ** 2], axis=1)
            betas[t] = np.linalg.lstsq(A, yN[itm] * np.exp(-r * dt * (N - t)), rcond=None)[0]
    tex = np.full(M, N)
    val = yN.copy()
    for t in range(N - 1, 0, -1):
        if nitm[t] == 0:
            continue
        cont = betas[t, 0] + betas[t, 1] * u[t] + betas[t, 2] * u[t] ** 2
        ex = (tex == N) & (pay[t] > 0) & (pay[t] > cont)
        tex[ex] = t
        val[ex] = pay[t, ex]
    cf = val * np.exp(-r * dt * (tex - 1))
    return dict(price=cf.mean(), n_exercised=int((tex < N).sum()), sum_nitm=int(nitm.sum()), tex=tex)


def test_fold_constants_and_table():
    S0, K, r, sig, T, N = 101.3, 97.0, 0.03, 0.27, 0.75, 91
    c0, g = orc.fold_constants(S0, K, r, sig, T, N)
    a = np.float32((r - 0.5 * sig * sig) * (T / N) * 1.4426950408889634074)
    assert c0 == float(np.float32(S0)) ** 2 / K
    assert g == 2.0 ** (2.0 * float(a))
    tab = orc.fold_table(N, c0, g)
    c = c0
    for t in range(N + 1):
        assert tab[t] == c
        c *= g
    # = S0^2 exp(2 drift t) / K up to the float32 rounding of the per-step exponent
    exact = S0 * S0 * np.exp(2 * (r - 0.5 * sig * sig) * (T / N) * np.arange(N + 1)) / K
    assert np.abs(tab / exact - 1).max() < 2e-6


@pytest.mark.parametrize("is_put,K,sig,N,P", [(1, 100.0, 0.2, 50, 10_000), (0, 95.0, 0.3, 20, 6_001), (1, 110.0, 0.4, 9, 2_000)])
def test_folded_oracle_against_numpy_on_the_explicit_matrix(is_put, K, sig, N, P):
    S0, r, T = 100.0, 0.05, 1.0
    half = orc.gbm_paths(P, N, S0, r, sig, T, seed=17, stream=2, antithetic=0)
    c0, g = orc.fold_constants(S0, K, r, sig, T, N)
    f = orc.lsm_two_pass_folded(half, K, r, T, is_put, c0, g)
    cK = orc.fold_table(N, c0, g)
    S64 = np.concatenate([half.astype(np.float64), cK[:, None] * K / half.astype(np.float64)], axis=1)
    ref = numpy_two_pass(S64, K, r, T, is_put)
    assert f["n_paths"] == 2 * P
    assert f["sum_nitm"] == ref["sum_nitm"]  # partner in the money <=> -K u' > 0 <=> C / S on the payoff's side of K
    assert f["price"] == pytest.approx(ref["price"], rel=1e-7)
    both = np.concatenate([f["texa"], f["texb"]])
    assert (both != ref["tex"]).sum() <= 2  # decisions: identical up to ties between two solvers' roundings
    assert abs(f["n_exercised"] - ref["n_exercised"]) <= 2


@pytest.mark.parametrize("is_put", [1, 0])
def test_folded_oracle_against_the_full_matrix_oracle(is_put):
    S0, K, r, sig, T, N, M = 100.0, 100.0, 0.05, 0.2, 1.0, 50, 40_000
    S = orc.gbm_paths(M, N, S0, r, sig, T, seed=7)
    full = orc.lsm_poly(S, K, r, T, is_put, "two_pass")
    c0, g = orc.fold_constants(S0, K, r, sig, T, N)
    f = orc.lsm_two_pass_folded(S[:, :M // 2], K, r, T, is_put, c0, g)
    assert f["price"] == pytest.approx(full["price"], rel=5e-6)
    assert abs(f["n_exercised"] - full["n_exercised"]) <= 4
    assert abs(f["sum_nitm"] - full["sum_nitm"]) <= 8
    assert (np.abs(f["betas"][1:N] - full["betas"][1:N]) <= 1e-3 * (1 + np.abs(full["betas"][1:N]))).all()


def test_folded_oracle_edges():
    S0, K, r, sig, T = 100.0, 105.0, 0.05, 0.2, 1.0
    for N, P in ((1, 512), (2, 3), (3, 1)):
        half = orc.gbm_paths(P, N, S0, r, sig, T, seed=1, antithetic=0)
        c0, g = orc.fold_constants(S0, K, r, sig, T, N)
        f = orc.lsm_two_pass_folded(half, K, r, T, 1, c0, g)
        full = orc.lsm_poly(orc.gbm_paths(2 * P, N, S0, r, sig, T, seed=1), K, r, T, 1, "two_pass")
        assert f["price"] == pytest.approx(full["price"], rel=1e-4)
        assert f["n_paths"] == 2 * P
