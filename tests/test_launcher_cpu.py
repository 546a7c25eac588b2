"""CPU suite: the rank launcher behind `n_gpus = N` from a plain process (options_model_amd/launcher.py, api.py).
The workers here speak the protocol only (tests/helpers/fake_rank_worker.py); the real worker runs in
tests/test_gpu_facade_ranks.py.  Reference callers this serves: options_model_2_ui.py:8-11, 87-133."""
import ast
import os
import sys
import time

import pytest

from options_model_amd import api, launcher

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE = [sys.executable, os.path.join(HERE, "helpers", "fake_rank_worker.py")]


def _pool(n, **kw):
    return launcher.RankPool(n, worker_argv=FAKE, start_timeout_s=60, **kw)


def test_roundtrip_rank_environment_and_stray_output():
    with _pool(3, devices=[0, 0, 2]) as p:
        envs = [None] * 3
        r0 = p.call("env", {})
        assert r0["RANK"] == "0" and r0["WORLD_SIZE"] == "3" and r0["MASTER_ADDR"] == "127.0.0.1"
        assert r0["device"] == "0" and r0["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and len(r0["OMC_RDZV_NONCE"]) == 16
        assert p.call("noise", {}) == dict(rank=0)  # a non-JSON line on a worker's stdout is skipped
        assert p.call("anything", dict(a=1))["echo"] == dict(a=1)
        pids = [q.pid for q in p.procs]
    assert all(q.poll() is not None for q in p.procs)  # closed: every worker has exited
    assert len(set(pids)) == 3


def test_value_error_on_every_rank_is_a_value_error_and_the_pool_lives_on():
    with _pool(2) as p:
        with pytest.raises(ValueError, match="must be positive"):
            p.call("bad", {})
        assert p.call("x", {})["rank"] == 0


def test_a_dead_rank_fails_the_call_names_the_rank_and_closes_the_pool():
    p = _pool(3)
    with pytest.raises(launcher.RankError, match="rank 1 exited"):
        p.call("die", dict(rank=1), timeout_s=30)
    assert p._closed and all(q.poll() is not None for q in p.procs)  # exactly the started processes, all gone
    with pytest.raises(launcher.RankError, match="closed"):
        p.call("x", {})


def test_an_error_on_one_rank_does_not_wait_for_its_peers():
    """A rank that answers with an error has left the collective call; its peers may sit in a collective without a
    deadline.  The call fails at once with that rank's error and the pool is taken down."""
    p = _pool(3)
    t0 = time.monotonic()
    with pytest.raises(launcher.RankError, match="rank 2: RuntimeError: kernel launch failed"):
        p.call("fail_then_hang", dict(rank=2), timeout_s=120)
    assert time.monotonic() - t0 < 30 and all(q.poll() is not None for q in p.procs)


def test_a_rank_local_value_error_does_not_leave_the_peers_waiting(monkeypatch):
    """a ValueError is treated as 'raised on every rank before anything collective'.  When only ONE rank
    raises it (a library check that fails for its shard), the others have entered the collective and would wait for the
    600 s / 3600 s deadline.  After a short grace period the call fails, names who refused and who went ahead, and the
    pool is taken down."""
    monkeypatch.setattr(launcher, "VALUE_ERROR_GRACE_S", 1.0)
    p = _pool(3)
    t0 = time.monotonic()
    with pytest.raises(launcher.RankError, match=r"rank\(s\) \[1\] refused bad_on_one \(selected another number of rows\) "
                                                 r"while rank\(s\) \[0, 2\] went ahead"):
        p.call("bad_on_one", dict(rank=1), timeout_s=20)  # grace = max(1 s, 5 % of the call's timeout) = 1 s
    assert time.monotonic() - t0 < 15 and p._closed and all(q.poll() is not None for q in p.procs)
    # the grace grows with the call's own timeout (a first call's ranks can be seconds apart): 5 % of 100 s = 5 s here
    p = _pool(3)
    t0 = time.monotonic()
    with pytest.raises(launcher.RankError, match="went ahead"):
        p.call("bad_on_one", dict(rank=1), timeout_s=100)
    assert 4.5 < time.monotonic() - t0 < 40 and p._closed


def test_deadline_ends_a_stuck_rank():
    p = _pool(2)
    t0 = time.monotonic()
    with pytest.raises(launcher.RankError, match=r"rank\(s\) \[1\] did not answer"):
        p.call("sleep", dict(rank=1, seconds=120), timeout_s=1.5)
    assert time.monotonic() - t0 < 30 and all(q.poll() is not None for q in p.procs)


def test_a_rank_that_cannot_start_fails_the_constructor(monkeypatch):
    monkeypatch.setenv("FAKE_DIE_AT_START", "1")
    with pytest.raises(launcher.RankError, match="rank 1"):
        _pool(2)


def test_parent_never_replaces_itself(monkeypatch):
    """Ranks are CHILD processes; the calling process never runs os.exec* (on the GPU pool an exec from a process
    that has initialised the GPU takes the machine down)."""
    def boom(*a, **k):
        raise AssertionError("os.exec* called in the parent")
    for name in ("execv", "execve", "execvp", "execvpe", "execl", "execle", "execlp", "execlpe"):
        monkeypatch.setattr(os, name, boom)
    with _pool(2) as p:
        assert p.call("x", {})["rank"] == 0
    for mod in ("launcher.py", "api.py", "_rank_worker.py", "dist.py"):
        tree = ast.parse(open(os.path.join(os.path.dirname(launcher.__file__), mod)).read())
        names = {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute)} | \
                {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)}
        assert not {n for n in names if n.startswith("exec") and n != "executable"}, mod


def test_in_job_detection(monkeypatch):
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert api._in_job(2) is False
    monkeypatch.setenv("WORLD_SIZE", "2")  # WORLD_SIZE alone (no RANK): not a rank
    assert api._in_job(2) is False
    monkeypatch.setenv("RANK", "1")
    assert api._in_job(2) is True
    with pytest.raises(RuntimeError, match="inside a 2-rank job"):
        api._in_job(4)
    monkeypatch.setenv("WORLD_SIZE", "1")  # a one-rank "job" may start ranks of its own
    assert api._in_job(4) is False


class _FakePool:
    def __init__(self):
        self.calls = []

    def call(self, fn, kw, timeout_s=600.0):
        self.calls.append((fn, kw))
        return dict(price=6.5, stderr=0.01, std=7.0, zero_prob=0.4, n_paths=kw["n_paths"], n_exercised=3, sum_nitm=9,
                    model="gbm", semantics=kw.get("semantics", "two_pass"), option_type=kw["option_type"],
                    timings_ms=dict(total=1.0), info=dict(n_gpus=4, rank=0, transport="rccl-native"))


def test_facade_from_a_plain_process_goes_through_the_pool(monkeypatch):
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    fake, seen = _FakePool(), []
    monkeypatch.setattr(launcher, "pool", lambda n, devices=None, env=None: (seen.append((n, devices)), fake)[1])
    res = api.price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, semantics="per_step", seed=7, n_gpus=4)
    assert res.price == 6.5 and res.n_paths == 4000 and res.info["launched_ranks"] == 4 and seen == [(4, None)]
    fn, kw = fake.calls[0]
    assert fn == "price_american_option" and kw["semantics"] == "per_step" and kw["seed"] == 7 and "n_gpus" not in kw
    api.price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, regressor="nn", n_gpus=2, device=[3, 5],
                              nn_options=dict(nn_hidden=64, torch_seed=11))
    assert seen[-1] == (2, [3, 5]) and fake.calls[-1][0] == "price_american_option_nn"
    assert fake.calls[-1][1]["nn_hidden"] == 64 and fake.calls[-1][1]["torch_seed"] == 11
    # the reference's validation (options_model_3.py:447-452) happens in the parent, before any rank is asked
    n = len(fake.calls)
    with pytest.raises(ValueError, match="S0, K, T must be positive"):
        api.price_american_option(-1.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, n_gpus=4)
    with pytest.raises(ValueError, match="do not pass ctx"):
        api.price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, n_gpus=4, ctx=object())
    # ONE card for several ranks is an argument error (real RCCL refuses it -- it used to surface as "the ranks did not
    # come up" after the start timeout); only the tests' shared-memory stand-in runs that way
    monkeypatch.delenv("OMC_RCCL_LIB", raising=False)
    with pytest.raises(ValueError, match="one device per rank"):
        api.price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, n_gpus=4, device=0)
    with pytest.raises(ValueError, match="lists 2 cards for n_gpus=4"):
        api.price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, n_gpus=4, device=[0, 1])
    assert len(fake.calls) == n
    monkeypatch.setenv("OMC_RCCL_LIB", "/somewhere/librccl_standin.so")
    api.price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 4000, 20, n_gpus=2, device=0)
    assert seen[-1] == (2, [0, 0])


def test_advanced_pricer_n_gpus_routes_through_the_facade(monkeypatch):
    from options_model_amd import AdvancedOptionPricer, RNGManager
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    fake = _FakePool()
    monkeypatch.setattr(launcher, "pool", lambda n, devices=None, env=None: fake)
    p = AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(42),
                             use_control_variate=False, regressor="poly", n_gpus=2)
    assert p.price_american_option(100.0, 1.0, 10000, 50) == 6.5
    fn, kw = fake.calls[0]
    sc_seeds = RNGManager(42)
    assert kw["seed"] == sc_seeds.get_child_seed() and kw["n_paths"] == 10000 and kw["n_steps"] == 50
    monkeypatch.setenv("OMC_N_GPUS", "4")  # the UI cannot pass the argument: the environment can
    assert AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, regressor="poly").n_gpus == 4


def test_omc_device_spreads_processes_over_gpus(monkeypatch):
    """OMC_DEVICE=auto: the reference's UIs fan a curve job's spot values over a spawn pool (options_model_2_ui.py:
    87-133); each worker process picks a GPU from its pid, so the pool uses the node without a change to the caller."""
    from options_model_amd import _ffi
    monkeypatch.delenv("OMC_DEVICE", raising=False)
    assert _ffi.resolve_device(None) == 0 and _ffi.resolve_device(3) == 3
    monkeypatch.setenv("OMC_DEVICE", "5")
    assert _ffi.resolve_device(None) == 5 and _ffi.resolve_device(1) == 1  # an explicit argument wins
    monkeypatch.setenv("OMC_DEVICE", "auto")
    monkeypatch.setattr(_ffi, "device_count", lambda: 8)
    monkeypatch.setattr(os, "getpid", lambda: 4243)
    assert _ffi.resolve_device(None) == 4243 % 8
    monkeypatch.setattr(_ffi, "device_count", lambda: 0)
    assert _ffi.resolve_device(None) == 0
