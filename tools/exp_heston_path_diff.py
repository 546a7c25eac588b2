#!/usr/bin/env python3
"""Device against C oracle path by path for one Heston case of the fuzz soak (full truncation, v0 = 0, xi = 1.5):
where does the worst path part ways?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from options_model_amd import _ffi
from oracle import cpu as orc
c = {'M': 1000, 'N': 100, 'S0': 142.22630343041476, 'r': 0.08, 'T': 1.6084565292525008, 'v0': 0.0, 'kappa': 0.6560452863927128,
     'theta': 0.2, 'xi': 1.5, 'rho': -0.7, 'scheme': 1, 'seed': 2052329983, 'stream': 0}
ctx = _ffi.Context(0)
Sd = ctx.heston_paths(c["M"], c["N"], c["S0"], c["r"], c["T"], c["v0"], c["kappa"], c["theta"], c["xi"], c["rho"], c["seed"], c["stream"], scheme=1)
G = Sd.to_host().astype(np.float64)
O = orc.heston_paths(c["M"], c["N"], c["S0"], c["r"], c["T"], c["v0"], c["kappa"], c["theta"], c["xi"], c["rho"], c["seed"], c["stream"], 0, 1).astype(np.float64)
rel = np.abs(G / O - 1)
print("max rel diff per step (first 12):", np.round(rel.max(axis=1)[:12], 9))
j = int(np.argmax(rel[-1]))
print("worst path", j, "terminal device", G[-1, j], "oracle", O[-1, j], "rel", rel[-1, j])
first = int(np.argmax(rel[:, j] > 1e-4)) if (rel[:, j] > 1e-4).any() else -1
print("first step with rel diff > 1e-4:", first)
for t in range(max(0, first - 3), min(c["N"] + 1, first + 4)):
    print(t, G[t, j], O[t, j], rel[t, j])
print("paths with terminal rel diff > 1e-4:", int((rel[-1] > 1e-4).sum()), " > 5e-5:", int((rel[-1] > 5e-5).sum()))
