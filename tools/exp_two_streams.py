#!/usr/bin/env python3
"""Do two independent sequences of pricings on two contexts (two streams, two sets of buffers) finish faster together
than one after the other?  (Kernels of one pricing leave the chip partly idle while they ramp up and drain.)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N, G = 252, 40
ctxs = [_ffi.Context(0) for _ in range(3)]
def seq(ctx, base, n=G):
    return ctx.price_american_seq([_ffi.make_params(semantics="two_pass", n_paths=M, n_steps=N, seed=42, stream=base + i) for i in range(n)])
for c in ctxs:
    seq(c, 0, 10)
t0 = time.perf_counter(); seq(ctxs[0], 100); seq(ctxs[0], 200); t1 = time.perf_counter() - t0
print(f"M={M}: one context, {2 * G} pricings back to back: {t1 / (2 * G) * 1e3:.4f} ms per pricing")
for k in (2, 3):
    th = [threading.Thread(target=seq, args=(ctxs[i], 100 * (i + 1), 2 * G // k * 1)) for i in range(k)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"M={M}: {k} contexts concurrently, {k * (2 * G // k)} pricings: {dt / (k * (2 * G // k)) * 1e3:.4f} ms per pricing")
