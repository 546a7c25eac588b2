#!/usr/bin/env python3
"""EXPERIMENT: do two sequences of pricings on two streams of one card finish sooner than one after the other?

Every kernel of the folded two-pass pricing leaves a pipe idle (generator: HBM writes, VALU 40 %; the sweeps: float64
issue 65 %, HBM 0.4) and ends in a tail, so kernels of DIFFERENT pricings could fill each other's gaps.  Two contexts
(own streams) price K pricings each -- first one after the other, then from two host threads at once (ctypes releases
the GIL in the library call).  Prints ms per pricing both ways.  usage: exp_two_streams.py [paths] [steps] [K] [contexts]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from options_model_amd import _ffi  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 252
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    ctxs = [_ffi.Context(0) for _ in range(C)]

    def run(c, base):
        ps = [_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=42, stream=base + i) for i in range(K)]
        return c.price_american_seq(ps)

    for i, c in enumerate(ctxs):  # warm: buffers, clocks
        run(c, 1000 * i)
        run(c, 1000 * i)
    t0 = time.perf_counter()
    serial = [run(c, 1000 * i) for i, c in enumerate(ctxs)]
    t_serial = time.perf_counter() - t0
    out = [None] * C
    th = [threading.Thread(target=lambda i=i: out.__setitem__(i, run(ctxs[i], 1000 * i))) for i in range(C)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_conc = time.perf_counter() - t0
    same = all(a["sum"] == b["sum"] and a["n_exercised"] == b["n_exercised"] for s, o in zip(serial, out) for a, b in zip(s, o))
    print(json.dumps({"paths": M, "steps": N, "pricings_per_context": K, "contexts": C, "folded": serial[0][0]["folded"],
                      "ms_per_pricing_one_after_the_other": 1e3 * t_serial / (C * K),
                      "ms_per_pricing_concurrent": 1e3 * t_conc / (C * K), "speedup": t_serial / t_conc, "bit_equal": same}))


if __name__ == "__main__":
    main()
