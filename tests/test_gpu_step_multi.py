"""K pricings per launch in the per-step flows (omc_price_american_seq, semantics reference / textbook).

North_star's per-timestep kernel is latency-bound for ONE pricing (13 MB per launch at 1M paths); a sequence of
pricings of one geometry therefore shares its launches: K path matrices, one launch per time step for all K.
The contract: every pricing of the sequence returns the BITS of its own omc_price_american call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("price", "sum", "sumsq", "n_exercised", "n_zero", "sum_nitm", "n_paths")


def _same(a, b):
    for k in KEYS:
        assert a[k] == b[k], (k, a[k], b[k])


@pytest.mark.parametrize("sem", ["reference", "textbook"])
@pytest.mark.parametrize("M,N,n", [
    (1_000_000, 20, 5),     # 245 slots, one chunk per thread; 5 pricings -> one batch of 5
    (300_000, 30, 11),      # ragged last slot; 11 pricings with K = 8 -> batches of 8 + 3
    (2_000_000, 6, 3),      # 256 slots, two chunks per thread and slot
    (30_002, 12, 4),        # n_paths % 4 != 0: scalar accesses
    (5_000, 1, 3),          # a single time step: only the initialising launch
    (64, 3, 16),            # tiny problem, 8 + 8 pricings
])
def test_k_pricings_per_launch_return_each_pricings_own_bits(ctx, sem, M, N, n):
    from options_model_amd import _ffi
    ps = [_ffi.make_params(semantics=sem, n_paths=M, n_steps=N, seed=11, stream=i, K=100.0 - (i % 3), is_put=(i % 4 != 3))
          for i in range(n)]
    ctx.set_option("seq_step_k", -1)
    multi = ctx.price_american_seq(ps)
    ctx.set_option("seq_step_k", 1)        # one pricing after the other (round 2's path)
    seq1 = ctx.price_american_seq(ps)
    ctx.set_option("seq_step_k", 32 if n > 8 else 3)   # another batch width: the summation tree does not depend on it
    multi3 = ctx.price_american_seq(ps)
    ctx.set_option("seq_step_k", -1)
    for p, a, b, c3 in zip(ps, multi, seq1, multi3):
        one = ctx.price_american(p)
        _same(a, one)
        _same(b, one)
        _same(c3, one)
    assert len({o["price"] for o in multi}) == n


def test_mixed_sequences_take_the_single_path(ctx):
    """Different geometry / flows in one sequence: nothing is batched, results unchanged."""
    from options_model_amd import _ffi
    ps = [_ffi.make_params(semantics="reference", n_paths=20_000, n_steps=20, seed=7, K=95.0),
          _ffi.make_params(semantics="reference", n_paths=20_000, n_steps=21, seed=7),
          _ffi.make_params(semantics="textbook", n_paths=20_000, n_steps=20, seed=3, T=0.5)]
    for p, s in zip(ps, ctx.price_american_seq(ps)):
        _same(s, ctx.price_american(p))


def test_heston_and_kept_state(ctx):
    """Heston paths through the shared launches; decisions equal the single call's (same fits)."""
    from options_model_amd import _ffi
    ps = [_ffi.make_params(model="heston", is_put=True, semantics="reference", n_paths=120_000, n_steps=25, seed=5,
                           stream=i, heston_scheme="full_truncation") for i in range(4)]
    for p, s in zip(ps, ctx.price_american_seq(ps)):
        _same(s, ctx.price_american(p))


def test_k_pricings_with_external_moments(ctx):
    """The multi-GPU form: per step the K moment vectors are reduced into ONE [K][8] block that goes through the
    all-reduce hook in one call of 8K doubles (identity / doubling hooks stand in for 1 and 2 equal ranks)."""
    import torch

    from options_model_amd import _ffi
    from options_model_amd.dist import _DevPtr
    stream = torch.cuda.Stream()
    c = _ffi.Context(0, stream=stream.cuda_stream)
    try:
        N, n = 15, 5
        ps = [_ffi.make_params(semantics="reference", n_paths=40_000, n_steps=N, seed=21, stream=i) for i in range(n)]
        base = [ctx.price_american(p) for p in ps]
        calls = []

        def ident(dptr, count):
            calls.append(count)
            torch.as_tensor(_DevPtr(dptr, count), device="cuda").add_(0.0)

        c.set_allreduce_hook(ident)
        with torch.cuda.stream(stream):
            out = c.price_american_seq(ps)
        # first the vote on K (34 doubles: every rank of a job takes part, whatever its own K -- ADVICE r4), then N - 1
        # collectives of 8K doubles (K = 5) + one of 8n result sums -- not n (N - 1) of 8
        assert calls[0] == 34 and calls.count(8 * n) == (N - 1) + 1 and len(calls) == N + 1
        for a, b in zip(out, base):
            _same(a, b)

        def doubling(dptr, count):
            torch.as_tensor(_DevPtr(dptr, count), device="cuda").mul_(2.0)

        c.set_allreduce_hook(doubling)
        c.set_option("world_size", 2)
        with torch.cuda.stream(stream):
            out2 = c.price_american_seq(ps)
        c.set_allreduce_hook(None)
        c.set_option("world_size", 1)
        for a, b in zip(out2, base):
            assert a["price"] == pytest.approx(b["price"], rel=1e-12)
            assert a["n_exercised"] == 2 * b["n_exercised"] and a["sum_nitm"] == 2 * b["sum_nitm"]
            assert a["n_paths"] == 2 * b["n_paths"]
    finally:
        c.close()


def test_large_shared_launch_against_the_oracle(ctx):
    """4 x 1M paths x 60 steps in shared launches; pricing 2 is checked against the CPU oracle on the same Philox
    stream (1e-3 relative, north_star's tolerance) -- the other three against their single calls, bit for bit."""
    from options_model_amd import _ffi
    from oracle import cpu as orc
    M, N = 1_000_000, 60
    ps = [_ffi.make_params(semantics="reference", n_paths=M, n_steps=N, seed=42, stream=100 + i) for i in range(4)]
    outs = ctx.price_american_seq(ps)
    for i in (0, 1, 3):
        _same(outs[i], ctx.price_american(ps[i]))
    S = orc.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, 42, 102)
    ref = orc.lsm_poly(S, 100.0, 0.05, 1.0, True, "reference")
    assert outs[2]["price"] == pytest.approx(ref["price"], rel=1e-3)
    assert abs(outs[2]["n_exercised"] - ref["n_exercised"]) <= 2e-4 * M
