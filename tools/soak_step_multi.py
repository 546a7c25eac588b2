"""Soak: many sequences of per-step pricings sharing their launches (ring buffers wrap, workspaces are reused); the same
sequence must return the same bits every time, and memory use must not grow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi
ctx = _ffi.Context(0)
M, N = 500_000, 100
ref = None
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < float(sys.argv[1]) if len(sys.argv) > 1 else 20.0:
    for sem, k in (("reference", 16), ("textbook", 8), ("reference", 5)):
        ps = [_ffi.make_params(semantics=sem, n_paths=M, n_steps=N, seed=7, stream=i) for i in range(k)]
        out = [(o["price"], o["sumsq"], o["n_exercised"], o["sum_nitm"]) for o in ctx.price_american_seq(ps)]
        key = (sem, k)
        if ref is None: ref = {}
        if key not in ref: ref[key] = out
        assert ref[key] == out, (n, key)
        n += 1
print(f"{n} sequences in {time.perf_counter() - t0:.1f} s: every repetition returned the same bits")
ctx.close()
