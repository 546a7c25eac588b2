#!/usr/bin/env python3
"""Where does one launch of lsm_step_kernel spend its time?  Runs the per-step reference sweep with the
time-stamping build of the kernel (omc_set_option "step_stamps") and prints, per stamp, the spread over
the 256 workgroups relative to the launch's first workgroup entry, plus the gap to the previous launch."""
import ctypes as C
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi  # noqa: E402

NAMES = ["entry", "partials in", "fit solved", "barrier passed", "rows in", "paths done", "block sums", "exit"]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    N = 252
    ctx = _ffi.Context(0)
    ctx.set_option("step_stamps", 1)
    p = _ffi.make_params(semantics="reference", n_paths=M, n_steps=N, seed=42)
    for _ in range(3):
        ctx.price_american(p)
    out = ctx.price_american(p)
    nblk = min(256, (M + 4095) // 4096)
    buf = np.zeros((N + 1, nblk, 8), np.uint64)
    _ffi._check(ctx.lib, ctx.lib.omc_debug_read(ctx.handle, buf.ctypes.data, buf.nbytes))
    ctx.close()
    t = buf.astype(np.float64) * 0.01  # 100 MHz ticks -> microseconds
    rows = []
    for li in range(5, N - 1):  # launch index: t_step = N - li ; skip the first launches and t = 1
        k = t[li]
        start = k[:, 0].min()
        prev_end = t[li - 1][:, 7].max()
        rows.append(np.concatenate([[start - prev_end], np.median(k - start, axis=0), (k - start).max(axis=0)]))
    r = np.median(np.array(rows), axis=0)
    print(json.dumps(dict(M=M, ms_lsm=out["ms_lsm"], us_per_step=1e3 * out["ms_lsm"] / N,
                          gap_prev_exit_to_entry_us=r[0],
                          median_over_blocks_us=dict(zip(NAMES, np.round(r[1:9], 2))),
                          max_over_blocks_us=dict(zip(NAMES, np.round(r[9:17], 2))))))
    print(f"per step {1e3 * out['ms_lsm'] / N:.2f} us ; gap (last exit of launch t+1 -> first entry of launch t) {r[0]:.2f} us")
    for i, n in enumerate(NAMES):
        print(f"  {n:15s} median {r[1 + i]:6.2f} us   latest workgroup {r[9 + i]:6.2f} us")


if __name__ == "__main__":
    main()
