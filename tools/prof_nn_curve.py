"""Workload for rocprofv3: a 40-point curve at 10k paths with the reference's default regressor (one SingleLSMNet
3 x 128 per point, all trained side by side: omc_mlp_train_epoch_batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import AdvancedOptionPricer, RNGManager
mk = lambda: AdvancedOptionPricer(K=100.0, r=0.05, sigma=0.2, option_type="put", rng_manager=RNGManager(3), use_control_variate=False)
mk().compute_curve_for_S0(100.0, 1, 2, 10_000, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
recs = mk().compute_curve_for_S0(100.0, 1, 40, 10_000, False)
torch.cuda.synchronize()
print("40-point NN curve seconds", time.perf_counter() - t0, "first", recs[0]["Option Value"], "last", recs[-1]["Option Value"])
