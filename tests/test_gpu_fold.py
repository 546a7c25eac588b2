"""Antithetic-FOLDED storage of the fused GBM two-pass pricing (option "fold_antithetic", default on; include/omc.h,
options_model_amd/csrc/omc_lsm_dev.h) against its oracle (oracle/omc_oracle.c: orc_lsm_two_pass_folded) and against the
full-matrix pricing.

What is compared, and how tightly:
  * the stored half of the matrix               bit-equal to the first half of the full antithetic matrix
  * the cK table                                bit-equal (N sequential IEEE products on both sides)
  * folded pricing vs the folded oracle on the SAME half matrix: regression-set sizes, exercised / zero counts
                                                identical; price rel <= 1e-9 (order of the float64 sums)
  * folded vs full-matrix pricing               rel <= 5e-6: the partner's spot is C_t / S_t in float64 instead of its
                                                own float32 product of N roundings (measured: <= 2e-6 per spot)
  * a sequence of folded pricings               res[i] bit-equal to the single call of p[i], whatever the neighbours
"""
import numpy as np
import pytest

from oracle import cpu as orc
from options_model_amd import _ffi

pytestmark = pytest.mark.gpu


@pytest.fixture
def fctx(ctx):
    ctx.set_option("fold_antithetic", 2)  # whatever the size (the default, 1, folds from 65,536 paths on)
    yield ctx
    ctx.set_option("fold_antithetic", 1)


def test_default_rule_folds_large_pricings_only(ctx):
    """Option value 1 (the default): pricings of at least 65,536 paths; smaller ones keep the full matrix and so the bits of
    their batched form (omc_price_american_batch prices its members on full storage)."""
    ctx.set_option("fold_antithetic", 1)
    small = [_ffi.make_params(semantics="two_pass", n_paths=m, n_steps=9, seed=8, stream=i) for i, m in enumerate((4_000, 65_534))]
    big = _ffi.make_params(semantics="two_pass", n_paths=65_536, n_steps=9, seed=8)
    assert [ctx.price_american(p)["folded"] for p in small] == [0, 0] and ctx.price_american(big)["folded"] == 1
    keys = ("price", "sum", "sumsq", "n_exercised", "n_zero", "sum_nitm")
    for one, b in zip([ctx.price_american(p) for p in small], ctx.price_american_batch(small)):
        assert [one[k] for k in keys] == [b[k] for k in keys]
    assert [r["folded"] for r in ctx.price_american_seq(small + [big])] == [0, 0, 1]


def _half(ctx, p):
    """the matrix the folded pricing stores: first partners only, same Philox counters"""
    return ctx.gbm_paths(p.n_paths // 2, p.n_steps, p.S0, p.r, p.sigma, p.T, p.seed, p.stream, p.pair_offset, antithetic=False).to_host()


def _oracle(ctx, p):
    c0, g = orc.fold_constants(p.S0, p.K, p.r, p.sigma, p.T, p.n_steps)
    return orc.lsm_two_pass_folded(_half(ctx, p), p.K, p.r, p.T, p.is_put, c0, g)


def test_first_partners_are_the_first_half_of_the_full_matrix(ctx):
    M, N = 20_004, 37
    full = ctx.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, seed=11, stream=3, pair_offset=12345).to_host()
    half = ctx.gbm_paths(M // 2, N, 100.0, 0.05, 0.2, 1.0, seed=11, stream=3, pair_offset=12345, antithetic=False).to_host()
    assert np.array_equal(full[:, :M // 2], half)
    # and the partner the sweeps reconstruct is the stored one up to float32 rounding of its N products
    c0, g = orc.fold_constants(100.0, 100.0, 0.05, 0.2, 1.0, N)
    cK = orc.fold_table(N, c0, g)
    rebuilt = cK[:, None] * 100.0 / half.astype(np.float64)
    assert np.abs(rebuilt / full[:, M // 2:] - 1).max() < 4e-6


CASES = [
    # n_paths, n_steps, is_put, S0, K, sigma
    (1_000_000, 252, True, 100.0, 100.0, 0.2),      # the headline configuration
    (200_000, 50, False, 100.0, 95.0, 0.3),
    (20_002, 50, True, 100.0, 100.0, 0.2),          # ragged: 10,001 stored columns (scalar loads)
    (20_008, 31, True, 36.0, 40.0, 0.4),            # 10,004 columns: vector loads, last tile partly filled
    (1_026, 7, True, 100.0, 110.0, 0.2),
    (2, 3, True, 100.0, 120.0, 0.2),                # one pair
    (4_096, 1, True, 100.0, 105.0, 0.2),            # one step: no regression at all
    (6_000, 2, False, 100.0, 90.0, 0.5),
    (2_048, 4094, True, 100.0, 100.0, 0.2),         # the maximum step count (pass 2's table of fits fills the LDS)
    (300_000, 40, False, 80.0, 100.0, 0.3),         # call far out of the money: the partner carries most rows
    (300_000, 40, True, 80.0, 100.0, 0.3),          # put deep in the money: both partners in the money in most lanes
    (4_194_312, 12, True, 100.0, 100.0, 0.25),      # 2,097,156 stored columns: pass 2 takes its 16-byte form from 2^21 on
]


@pytest.mark.parametrize("M,N,is_put,S0,K,sig", CASES)
def test_folded_pricing_equals_its_oracle(fctx, M, N, is_put, S0, K, sig):
    p = _ffi.make_params(semantics="two_pass", is_put=is_put, n_paths=M, n_steps=N, S0=S0, K=K, sigma=sig, seed=2024, stream=5)
    r = fctx.price_american(p)
    o = _oracle(fctx, p)
    assert r["folded"] == 1 and r["n_paths"] == M
    assert r["sum_nitm"] == o["sum_nitm"]
    assert r["n_exercised"] == o["n_exercised"]
    assert r["n_zero"] == o["n_zero"]
    assert r["price"] == pytest.approx(o["price"], rel=1e-9, abs=1e-300)
    assert r["sumsq"] == pytest.approx(o["sumsq"], rel=1e-9, abs=1e-300)


@pytest.mark.parametrize("M,N,is_put,S0,K,sig", CASES[:6])
def test_folded_and_full_storage_agree(fctx, M, N, is_put, S0, K, sig):
    p = _ffi.make_params(semantics="two_pass", is_put=is_put, n_paths=M, n_steps=N, S0=S0, K=K, sigma=sig, seed=99, stream=1)
    f = fctx.price_american(p)
    fctx.set_option("fold_antithetic", 0)
    u = fctx.price_american(p)
    assert (f["folded"], u["folded"]) == (1, 0)
    if M <= 200_000:
        # the full-matrix pricing against the oracle END TO END (its own libm paths: spots 2e-5 apart, prices 1e-5)
        S = orc.gbm_paths(M, N, S0, p.r, sig, p.T, 99, 1)
        o = orc.lsm_poly(S, K, p.r, p.T, is_put, "two_pass")
        assert u["price"] == pytest.approx(o["price"], rel=1e-5 if M >= 1000 else 1e-3, abs=1e-6)
    if M >= 1000:
        assert f["price"] == pytest.approx(u["price"], rel=5e-6)
        assert abs(f["n_exercised"] - u["n_exercised"]) <= max(4, M // 20_000)
        assert abs(f["sum_nitm"] - u["sum_nitm"]) <= max(8, M * N // 500_000)
    else:
        assert f["price"] == pytest.approx(u["price"], rel=1e-4, abs=1e-6)


def test_who_folds(fctx):
    """Only antithetic GBM in the two-pass flow, and only when the library owns the matrix."""
    M, N = 8_192, 20
    for kw, want in ((dict(semantics="two_pass"), 1), (dict(semantics="reference"), 0), (dict(semantics="textbook"), 0),
                     (dict(semantics="two_pass", antithetic=False), 0), (dict(semantics="two_pass", model="heston"), 0)):
        p = _ffi.make_params(n_paths=M, n_steps=N, **kw)
        assert fctx.price_american(p)["folded"] == want, kw
    p = _ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N)
    keep = fctx.empty((N + 1, M), np.float32)
    r = fctx.price_american(p, keep_paths=keep)
    assert r["folded"] == 0
    S = keep.to_host()
    assert np.array_equal(S, fctx.gbm_paths(M, N, p.S0, p.r, p.sigma, p.T, p.seed).to_host())
    o = orc.lsm_poly(S, p.K, p.r, p.T, 1, "two_pass")
    assert (r["n_exercised"], r["sum_nitm"]) == (o["n_exercised"], o["sum_nitm"])
    keep.free()


def test_a_sequence_of_folded_pricings_keeps_every_single_calls_bits(fctx):
    """Strikes, spots, volatilities and payoff sides change from one pricing to the next: every change refills the cK
    table on the stream, between two pricings that are both in flight."""
    M, N = 50_000, 40
    ps = [_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, K=k, S0=s0, sigma=sg, is_put=put, seed=5, stream=i)
          for i, (k, s0, sg, put) in enumerate([(100, 100, 0.2, True), (100, 100, 0.2, True), (105, 100, 0.2, True),
                                                (105, 98, 0.2, False), (95, 98, 0.35, True), (100, 100, 0.2, True)])]
    ps.insert(3, _ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, antithetic=False, seed=5, stream=77))  # not folded
    seq = fctx.price_american_seq(ps)
    keys = ("price", "sum", "sumsq", "n_exercised", "n_zero", "sum_nitm", "folded")
    for p, r in zip(ps, seq):
        one = fctx.price_american(p)
        assert [r[k] for k in keys] == [one[k] for k in keys]
    assert [r["folded"] for r in seq] == [1, 1, 1, 0, 1, 1, 1]
    o = _oracle(fctx, ps[5])
    assert seq[5]["n_exercised"] == o["n_exercised"] and seq[5]["price"] == pytest.approx(o["price"], rel=1e-9)


def test_folded_shards_add_up(fctx):
    """Two 'ranks' (pair_offset) of a folded pricing draw the pairs of the one-GPU pricing: their stored halves are its
    stored half, column for column."""
    M, N = 40_000, 12
    p = _ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=3)
    whole = _half(fctx, p)
    a = _ffi.make_params(semantics="two_pass", n_paths=M // 2, n_steps=N, seed=3, pair_offset=0)
    b = _ffi.make_params(semantics="two_pass", n_paths=M // 2, n_steps=N, seed=3, pair_offset=M // 4)
    assert np.array_equal(np.concatenate([_half(fctx, a), _half(fctx, b)], axis=1), whole)
