"""The NN regressor sharded over ranks (options_model_3.py:542-651) on the GPU: rank processes started
by the facade (launcher.RankPool), sharing this box's one card through the shared-memory librccl stand-in, train ONE
network on the rows of all shards -- statistics, minibatch composition, dropout masks and pass-2 masks are those of the
single-GPU run, so the sharded job must reproduce it up to float32 summation order of the gradient sums."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ARGS = (100.0, 100.0, 0.05, 0.2, 1.0, 120_000, 40)
NN = dict(nn_hidden=64, nn_layers=2, nn_epochs=4, nn_lr=1e-3, torch_seed=77)


@pytest.fixture(scope="module")
def standin():
    import rccl_standin
    return rccl_standin.build()


@pytest.fixture()
def plain_process(monkeypatch, standin):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("OMC_RCCL_LIB", standin)
    from options_model_amd import launcher
    yield launcher
    launcher.close_pools()


@pytest.fixture(scope="module")
def single():
    """The single-GPU run: with and without dropout (training + inference)."""
    from options_model_amd import price_american_option
    out = {}
    for tag, drop in (("drop", 0.1), ("nodrop", 0.0)):
        out[tag] = price_american_option(*ARGS, regressor="nn", seed=11, nn_options=dict(NN, nn_dropout=drop))
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_nn_regressor_equals_the_single_gpu_run(single, plain_process, world):
    kw = dict(S0=ARGS[0], K=ARGS[1], r=ARGS[2], sigma=ARGS[3], T=ARGS[4], n_paths=ARGS[5], n_steps=ARGS[6], model="GBM",
              option_type="put", heston_params=None, seed=11, stream=0)
    pool = plain_process.pool(world, [0] * world)
    for tag, drop in (("nodrop", 0.0), ("drop", 0.1)):
        res = pool.call_all("price_american_option_nn", dict(kw, **NN, nn_dropout=drop), timeout_s=600)
        one = single[tag]
        # every rank: the same weights (bit for bit), the same global result
        assert len({r["info"]["params_checksum"] for r in res}) == 1
        assert len({r["price"] for r in res}) == 1 and len({r["info"]["best_loss"] for r in res}) == 1
        r0 = res[0]
        print(f"world {world} {tag}: sharded price {r0['price']:.6f} loss {r0['info']['best_loss']:.6f} | single "
              f"{one.price:.6f} loss {one.info['best_loss']:.6f}")
        assert r0["sum_nitm"] == one.sum_nitm and r0["n_paths"] == one.n_paths       # the same rows
        assert r0["info"]["batch"] == one.info["batch"] and r0["info"]["optimizer_steps"] == one.info["optimizer_steps"]
        # the same minibatches and masks; per-rank float32 partial sums added in another association (in float64):
        # (measured on MI355X: loss 3e-6 / 4e-5 relative without / with dropout; price 2e-4 .. 2.6e-3 relative)
        assert r0["info"]["best_loss"] == pytest.approx(one.info["best_loss"], rel=2e-4)
        # the weights agree to float32 rounding amplified by ~2,200 Adam steps; paths whose payoff sits that close to
        # the network's output exercise at another step.  Bound: 1.5 standard errors of the single-GPU price itself.
        assert abs(r0["price"] - one.price) <= 1.5 * one.stderr, (r0["price"], one.price, one.stderr)


def test_facade_nn_n_gpus_from_a_plain_process(single, plain_process):
    from options_model_amd import price_american_option
    res = price_american_option(*ARGS, regressor="nn", seed=11, n_gpus=2, device=0, nn_options=dict(NN, nn_dropout=0.0))
    assert res.info["launched_ranks"] == 2 and res.info["trainer"] == "hip" and res.info["transport"] == "rccl-native"
    assert abs(res.price - single["nodrop"].price) <= 1.5 * single["nodrop"].stderr and res.sum_nitm == single["nodrop"].sum_nitm
    with pytest.raises(ValueError, match="multiple of"):
        price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 120_002, 40, regressor="nn", n_gpus=2, device=0)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_nn_tiny_and_empty_shards(plain_process, world):
    """Edge cases of the sharded flow: a handful of rows (one ragged minibatch per epoch, ranks owning a few rows or
    NONE of a step's minibatch -- empty launches are skipped, the rank still takes part in every all-reduce), a rank
    without any row, and a job without any row (options_model_3.py:518-519: fall back to the discounted terminal
    payoff).  Each equals the single-GPU call on the same seeds."""
    from options_model_amd import price_american_option
    nn = dict(nn_hidden=64, nn_layers=2, nn_epochs=3, nn_dropout=0.0, torch_seed=5)
    cases = [
        dict(S0=100.0, K=100.0, T=1.0, n_paths=128, n_steps=6, option_type="put"),      # ~300 rows, one minibatch
        dict(S0=118.0, K=100.0, T=0.25, n_paths=16 * world, n_steps=4, option_type="put"),  # a few rows; some ranks none
        dict(S0=50.0, K=100.0, T=0.5, n_paths=64, n_steps=5, option_type="call"),        # nothing ever in the money
        dict(S0=100.0, K=100.0, T=1.0, n_paths=3000, n_steps=9, option_type="put"),      # several minibatches of 256
        dict(S0=100.0, K=100.0, T=1.0, n_paths=2048, n_steps=8, option_type="put", model="Heston"),  # Heston shards
    ]
    pool = plain_process.pool(world, [0] * world)
    for c in cases:
        model = c.get("model", "GBM")
        kw = dict(S0=c["S0"], K=c["K"], r=0.05, sigma=0.2, T=c["T"], n_paths=c["n_paths"], n_steps=c["n_steps"],
                  model=model, option_type=c["option_type"], heston_params=None, seed=3, stream=1)
        res = pool.call_all("price_american_option_nn", dict(kw, **nn), timeout_s=300)
        one = price_american_option(c["S0"], c["K"], 0.05, 0.2, c["T"], c["n_paths"], c["n_steps"], model=model,
                                    option_type=c["option_type"], regressor="nn", seed=3, stream=1, nn_options=nn)
        assert len({r["price"] for r in res}) == 1, c
        r0 = res[0]
        assert r0["sum_nitm"] == one.sum_nitm and r0["n_paths"] == one.n_paths, (c, r0["sum_nitm"], one.sum_nitm)
        assert r0["price"] == pytest.approx(one.price, rel=2e-3, abs=1e-9), (c, r0["price"], one.price)
        if one.sum_nitm:
            assert r0["info"]["optimizer_steps"] == one.info["optimizer_steps"]
            assert r0["info"]["best_loss"] == pytest.approx(one.info["best_loss"], rel=1e-3)
