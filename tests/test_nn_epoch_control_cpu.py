"""The host side of the NN flow's training (options_model_amd.nn_regressor.EpochControl) pinned to the reference:

1. traces RECORDED from real runs of the reference's price_american_enhanced_lsm (tools/capture_golden_epochs.py ->
   tests/golden/nn_epoch_trace.npz: per-epoch mean loss, learning rate after scheduler.step, the epoch whose weights
   it restored, the epoch it stopped at; options_model_3/options_model_3.py:574-615) are the script: fed the recorded
   losses, EpochControl must produce the recorded learning rates, stop where the reference stopped and keep the
   weights of the epoch the reference restored;
2. scripted loss sequences against torch.optim.lr_scheduler.ReduceLROnPlateau(patience=5, factor=0.5, min_lr=1e-6)
   itself plus a literal transcription of the reference's `avg_loss < best - 1e-6` / patience-8 rule.
"""
import os

import numpy as np
import pytest

from options_model_amd.nn_regressor import EpochControl

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nn_epoch_trace.npz")


@pytest.mark.parametrize("tag", ["g4", "noisy_a", "noisy_b", "short"])
def test_recorded_reference_traces_are_reproduced(tag):
    g = np.load(GOLDEN)
    losses, lrs = g[f"{tag}_losses"], g[f"{tag}_lrs"]
    M, N, hidden, epochs, lr, seed = g[f"{tag}_config"]
    ctl = EpochControl(lr)
    stopped_at = None
    for e, loss in enumerate(losses):
        keep, stop = ctl.step(loss)
        assert ctl.lr == lrs[e], (tag, e)          # the very doubles torch's scheduler produced
        if stop:
            stopped_at = e + 1
            break
    epochs_run = stopped_at if stopped_at is not None else len(losses)
    assert epochs_run == int(g[f"{tag}_epochs_run"]) == len(losses)
    assert (stopped_at is not None) == (len(losses) < int(epochs))   # early stop iff the reference stopped early
    assert ctl.best_epoch == int(g[f"{tag}_best_epoch"])              # the weights the reference restored
    assert ctl.best_loss == losses[ctl.best_epoch]


def test_traces_exercise_what_they_should():
    g = np.load(GOLDEN)
    assert g["g4_lrs"][-1] < g["g4_lrs"][0]                         # the scheduler fired in the G4 run
    assert int(g["noisy_a_epochs_run"]) < 120 and int(g["noisy_b_epochs_run"]) < 200   # real early stops
    assert int(g["noisy_a_best_epoch"]) < int(g["noisy_a_epochs_run"]) - 1


def _reference_rule(losses, lr0, max_epochs):
    """Literal restatement of :578-611 around torch's own scheduler (an SGD optimizer over a dummy parameter only
    carries the learning rate).  -> (lrs, epochs_run, best_epoch)"""
    import torch
    holder = torch.zeros(1, requires_grad=True)
    opt = torch.optim.SGD([holder], lr=lr0)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=5, factor=0.5, min_lr=1e-6)
    best_loss, best_epoch, patience_counter = float("inf"), -1, 0
    lrs = []
    epoch = -1
    for epoch in range(min(max_epochs, len(losses))):
        avg_loss = float(losses[epoch])
        scheduler.step(avg_loss)
        lrs.append(opt.param_groups[0]["lr"])
        if avg_loss < best_loss - 1e-6:
            best_loss, best_epoch, patience_counter = avg_loss, epoch, 0
        else:
            patience_counter += 1
            if patience_counter >= 8:
                break
    return lrs, epoch + 1, best_epoch


@pytest.mark.parametrize("seed", range(12))
def test_scripted_losses_against_torch_scheduler(seed):
    rng = np.random.default_rng(seed)
    n = 80
    kind = seed % 4
    if kind == 0:      # steady decay with noise, then a plateau
        losses = 1.0 * np.exp(-0.1 * np.arange(n)) + 0.3 + 0.01 * rng.standard_normal(n)
    elif kind == 1:    # plateau from the start: repeated halvings down to min_lr
        losses = 0.5 + 1e-5 * rng.standard_normal(n)
    elif kind == 2:    # improvements smaller than the scheduler's relative threshold but larger than 1e-6
        losses = 0.7 - 2e-5 * np.arange(n)
    else:              # saw-tooth
        losses = 0.6 + 0.05 * ((np.arange(n) % 7) == 0) - 0.002 * np.arange(n) + 0.004 * rng.standard_normal(n)
    lr0 = [1e-3, 5e-2, 3e-6, 1e-3][kind]   # (3e-6: runs into min_lr and the eps rule)
    want_lrs, want_epochs, want_best = _reference_rule(losses, lr0, n)
    ctl = EpochControl(lr0)
    got_lrs, epochs_run = [], 0
    for e in range(n):
        keep, stop = ctl.step(losses[e])
        got_lrs.append(ctl.lr)
        epochs_run = e + 1
        if stop:
            break
    assert epochs_run == want_epochs and ctl.best_epoch == want_best
    assert got_lrs == want_lrs


def test_min_lr_and_eps_rule():
    ctl = EpochControl(1.5e-6)
    for _ in range(7):
        ctl.step(1.0)
    assert ctl.lr == 1e-6                   # max(0.75e-6, min_lr)
    for _ in range(30):
        ctl.step(1.0)
    assert ctl.lr == 1e-6                   # 1e-6 -> 1e-6 would change nothing (eps 1e-8): unchanged
