#!/usr/bin/env python3
"""Per-step timeline of the persistent sweep (omc_lsm_persist.hip) from its in-kernel stamps."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi  # noqa: E402

NAMES = ["loop top", "gather done", "(polls)", "fit solved", "barrier 2", "apply done", "moments done", "published"]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    N = 252
    ctx = _ffi.Context(0)
    ctx.set_option("step_stamps", 1)
    ctx.set_option("step_persistent", 1)
    p = _ffi.make_params(semantics="reference", n_paths=M, n_steps=N, seed=42)
    for _ in range(3):
        ctx.price_american(p)
    out = ctx.price_american(p)
    nblk = min(256, (M + 4095) // 4096)
    buf = np.zeros((N + 1, nblk, 8), np.uint64)
    _ffi._check(ctx.lib, ctx.lib.omc_debug_read(ctx.handle, buf.ctypes.data, buf.nbytes))
    ctx.close()
    t = buf.astype(np.float64)
    rows = []
    for li in range(5, N - 2):
        k = t[li].copy()
        polls = k[:, 2].copy()
        k *= 0.01
        nxt = t[li + 1][:, 0] * 0.01
        rel = k - k[:, 0:1]
        rows.append(np.concatenate([np.median(rel, axis=0), [np.median(polls), polls.max(), np.median(nxt - k[:, 0])]]))
    r = np.median(np.array(rows), axis=0)
    print(f"M={M}: ms_lsm {out['ms_lsm']:.3f} -> {1e3 * out['ms_lsm'] / N:.2f} us per step; loop period (median block) {r[10]:.2f} us; "
          f"polls median {r[8]:.0f} max {r[9]:.0f}")
    for i, n in enumerate(NAMES):
        if i != 2:
            print(f"  {n:14s} +{r[i]:6.2f} us")


if __name__ == "__main__":
    main()
