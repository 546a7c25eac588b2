#!/usr/bin/env python3
"""Seed-to-seed spread of the REFERENCE's own 10k x 50 NN pricing (build container only).
Gives the statistical band the NN end-to-end GPU test is allowed (tests/test_gpu_nn.py):
the reference's answer moves by several stderr from seed to seed because the trained net,
the dropout-at-inference noise and the look-ahead rule (F2) all feed the price.
Appends to tests/golden/scalars.json["reference_nn_seed_band"] (the reference's default net, 3 x 128)
or, with --hidden 64 first on the command line, ["reference_nn_seed_band_h64"] (nn_hidden=64: the 3 x 64
net that this repo trains and applies entirely with its own kernels)."""
import json, os, sys, time, types
sys.modules.setdefault("yfinance", types.ModuleType("yfinance"))
sys.path.insert(0, "/root/reference/options_model_3")
import torch
torch.set_num_threads(int(os.environ.get("REF_THREADS", "8")))
import options_model_3 as om  # noqa: E402

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scalars.json")
sc = json.load(open(out))
args = sys.argv[1:]
hidden = 128
if args[:1] == ["--hidden"]:
    hidden, args = int(args[1]), args[2:]
key = "reference_nn_seed_band" if hidden == 128 else f"reference_nn_seed_band_h{hidden}"
band = sc.get(key, {})
for seed in [int(s) for s in args] or [1, 2, 3]:
    if str(seed) in band:
        continue
    t0 = time.time()
    p = om.AdvancedOptionPricer(K=100, r=0.05, sigma=0.2, option_type="put",
                                rng_manager=om.RNGManager(seed), use_control_variate=False, nn_hidden=hidden)
    band[str(seed)] = float(p.price_american_option(100.0, 1.0, 10000, 50))
    print(seed, band[str(seed)], f"{time.time()-t0:.0f}s", flush=True)
    sc[key] = band
    json.dump(sc, open(out, "w"), indent=1, sort_keys=True)
