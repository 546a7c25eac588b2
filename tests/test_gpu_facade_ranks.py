"""`n_gpus = N` from a PLAIN process: the facade and AdvancedOptionPricer start their own rank
processes (launcher.RankPool -> options_model_amd._rank_worker), one per GPU, and return the global result.  Here the
ranks share this box's one GPU through the shared-memory librccl stand-in (tests/rccl_standin), 2 and 4 of them (a GPU
box admits at most 6 processes on its card; pytest is one).  Reference callers this serves: the single-process
Streamlit script options_model_2_ui.py:87-133 and the spawn pool of options_model_3.py:1043-1056."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def standin():
    import rccl_standin
    return rccl_standin.build()


@pytest.fixture()
def plain_process(monkeypatch, standin):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("OMC_RCCL_LIB", standin)  # inherited by the rank processes
    from options_model_amd import launcher
    yield launcher
    launcher.close_pools()


@pytest.mark.parametrize("world", [2, 4])
def test_facade_starts_its_own_ranks_and_equals_the_unsharded_pricing(ctx, plain_process, world):
    from options_model_amd import price_american_option
    for sem in ("two_pass", "per_step", "textbook"):
        r = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, semantics=sem, seed=5, stream=2,
                                  n_gpus=world, device=0)
        one = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, semantics=sem, seed=5, stream=2, ctx=ctx)
        assert r.info["launched_ranks"] == world and r.info["transport"] == "rccl-native" and r.n_paths == 200_000
        assert r.price == pytest.approx(one.price, rel=1e-12) and r.stderr == pytest.approx(one.stderr, rel=1e-9)
        assert (r.n_exercised, r.sum_nitm) == (one.n_exercised, one.sum_nitm)
    pool = plain_process.pool(world, [0] * world)
    pids = [p.pid for p in pool.procs]
    h = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, model="Heston", option_type="call",
                              heston_scheme="full_truncation", seed=5, n_gpus=world, device=0)
    one = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, model="Heston", option_type="call",
                                heston_scheme="full_truncation", seed=5, ctx=ctx)
    assert h.price == pytest.approx(one.price, rel=1e-12)
    assert [p.pid for p in plain_process.pool(world, [0] * world).procs] == pids  # the same ranks served every call
    with pytest.raises(ValueError, match="multiple of"):  # raised on every rank before anything collective: pool lives
        price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_002, 40, n_gpus=world, device=0)
    again = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 200_000, 40, semantics="two_pass", seed=5, stream=2,
                                  n_gpus=world, device=0)
    assert again.n_paths == 200_000 and [p.pid for p in plain_process.pool(world, [0] * world).procs] == pids


@pytest.mark.parametrize("world", [2, 4])
def test_regressor_ols7_shards_its_fit_over_the_ranks(ctx, plain_process, world):
    """regressor="ols7" with n_gpus = N from a plain process: every rank sweeps its own paths, the ranks' (n, mean, co-moment)
    triples are merged by two small all-reduces, every rank solves the same 6 x 6 system -- the fit and the price of the
    unsharded pricing (same Philox pairs; sums in another order)."""
    from options_model_amd import price_american_option
    for kw in (dict(option_type="put"), dict(option_type="call", model="Heston", heston_scheme="full_truncation")):
        r = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 120_000, 30, regressor="ols7", seed=9, stream=1, n_gpus=world,
                                  device=0, **kw)
        one = price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 120_000, 30, regressor="ols7", seed=9, stream=1, ctx=ctx, **kw)
        assert r.info["launched_ranks"] == world and r.n_paths == 120_000 and r.sum_nitm == one.sum_nitm
        assert r.info["weights"] == pytest.approx(one.info["weights"], rel=1e-6, abs=1e-9)
        assert r.n_exercised == one.n_exercised and r.price == pytest.approx(one.price, rel=1e-12)


def test_advanced_pricer_n_gpus_from_a_plain_process(ctx, plain_process, golden):
    """The v3 class with the reference's own arguments plus n_gpus: same child seeds, same price as one GPU."""
    from options_model_amd import AdvancedOptionPricer, RNGManager
    kw = dict(K=100, r=0.05, sigma=0.2, option_type="put", use_control_variate=False, regressor="poly")
    two = AdvancedOptionPricer(rng_manager=RNGManager(42), n_gpus=2, devices=[0, 0], **kw)
    one = AdvancedOptionPricer(rng_manager=RNGManager(42), **kw)
    a, b = two.price_american_option(100.0, 1.0, 10000, 50), one.price_american_option(100.0, 1.0, 10000, 50)
    assert a == pytest.approx(b, rel=1e-12) and two.last_result["sum_nitm"] == one.last_result["sum_nitm"]
    assert two.rng_manager.get_child_seed() == golden["scalars"]["rng_manager_42_child_seeds"][2]
