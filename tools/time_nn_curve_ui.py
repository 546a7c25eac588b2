"""The Streamlit UI's default job for ONE spot value with the v3 drop-in's default regressor: 180 expiries
(intervals_per_day 2, total_points 180), 10k paths, one SingleLSMNet 3 x 128 per point, control variate on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import compute_curve_worker_enhanced
compute_curve_worker_enhanced(100.0, 100.0, 0.05, 0.2, "put", 1, 2, 3, 4000, False, False, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
recs = compute_curve_worker_enhanced(100.0, 100.0, 0.05, 0.2, "put", 42, 2, 180, 10_000, False, False, None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"180-point NN curve (10k paths, 3x128, CV on): {dt:.2f} s; first {recs[0]}, last {recs[-1]}; GB allocated peak {torch.cuda.max_memory_allocated()/1e9:.2f}")
