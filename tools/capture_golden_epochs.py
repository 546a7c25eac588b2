#!/usr/bin/env python3
"""Record the reference's own training-loop TRACE: per-epoch mean loss, learning rate, which epoch's weights it
restored and when it stopped (options_model_3/options_model_3.py:574-615).

Runs ONLY in the build container (needs /root/reference).  The REAL AdvancedOptionPricer.price_american_enhanced_lsm
runs unmodified; two observation hooks record numbers:
  * torch's ReduceLROnPlateau is replaced, for the duration of the run, by a subclass whose step() notes the metric
    it was given (= the reference's avg_loss), the optimizer's learning rate after the step, and a hash of the
    network's weights at that moment (the optimizer holds the parameters);
  * SingleLSMNet remembers its instance, so the weights the reference ends up with (after "Restore best weights")
    can be matched to the epoch they came from.
Output: tests/golden/nn_epoch_trace.npz -- float / int arrays only.  The CPU suite feeds `losses` to
options_model_amd.nn_regressor.EpochControl and must get back `lrs`, `epochs_run`, `best_epoch`
(tests/test_nn_epoch_control_cpu.py).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden_epochs.py
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_golden import OUT, import_reference  # noqa: E402


def state_hash(params):
    h = hashlib.sha256()
    for p in params:
        h.update(p.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def run(om, tag, out, M, N, hidden, epochs, lr, seed, option_type="put"):
    import torch
    trace = []
    nets = []
    OrigSched = torch.optim.lr_scheduler.ReduceLROnPlateau
    OrigNet = om.SingleLSMNet

    class Rec(OrigSched):
        def step(self, metrics, *a, **k):
            r = super().step(metrics, *a, **k)
            group = self.optimizer.param_groups[0]
            trace.append((float(metrics), float(group["lr"]), state_hash(group["params"])))
            return r

    class Spy(OrigNet):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            nets.append(self)

    torch.optim.lr_scheduler.ReduceLROnPlateau = Rec
    om.SingleLSMNet = Spy
    try:
        pricer = om.AdvancedOptionPricer(K=100.0, r=0.05, sigma=0.2, option_type=option_type,
                                         rng_manager=om.RNGManager(seed), nn_hidden=hidden, nn_epochs=epochs,
                                         nn_lr=lr, use_control_variate=False)
        price = pricer.price_american_option(100.0, 1.0, M, N)
    finally:
        torch.optim.lr_scheduler.ReduceLROnPlateau = OrigSched
        om.SingleLSMNet = OrigNet
    losses = np.array([t[0] for t in trace])
    lrs = np.array([t[1] for t in trace])
    final = state_hash(list(nets[0].parameters()))
    hashes = [t[2] for t in trace]
    # the reference restores the weights of its best epoch: find that epoch among the recorded ones
    best_epoch = max(i for i, h in enumerate(hashes) if h == final) if final in hashes else -1
    assert best_epoch >= 0, "the final weights are not those of any epoch end"
    out[f"{tag}_losses"] = losses
    out[f"{tag}_lrs"] = lrs
    out[f"{tag}_epochs_run"] = np.int64(len(trace))
    out[f"{tag}_best_epoch"] = np.int64(best_epoch)
    out[f"{tag}_config"] = np.array([M, N, hidden, epochs, lr, seed], dtype=np.float64)
    out[f"{tag}_price"] = np.float64(price)
    print(f"[{tag}] epochs_run={len(trace)}/{epochs} best_epoch={best_epoch} lr {lrs[0]:g} -> {lrs[-1]:g} "
          f"loss {losses[0]:.4f} -> {losses.min():.4f} price={price:.6f}")


def main():
    om = import_reference()
    import torch
    torch.set_num_threads(8)
    out = {}
    # the G4 configuration of capture_golden.py (the reference's defaults at fixture size)
    run(om, "g4", out, M=1024, N=50, hidden=128, epochs=25, lr=1e-3, seed=42)
    # long runs on little data at a high learning rate: noisy epoch losses -> the scheduler halves the rate
    # several times and the patience-8 rule ends training early
    run(om, "noisy_a", out, M=256, N=20, hidden=32, epochs=120, lr=2e-2, seed=7)
    run(om, "noisy_b", out, M=128, N=10, hidden=16, epochs=200, lr=5e-2, seed=11, option_type="call")
    run(om, "short", out, M=512, N=20, hidden=64, epochs=6, lr=1e-3, seed=3)
    np.savez_compressed(os.path.join(OUT, "nn_epoch_trace.npz"), **out)
    print("wrote", os.path.join(OUT, "nn_epoch_trace.npz"))


if __name__ == "__main__":
    main()
