"""Experiment: the per-step sweep of K pricings split over several streams (contexts), each stream's launches
confined to a share of the chip's workgroups, so that one stream's launch boundary / cold start overlaps the other
streams' streaming phase.  Prints ms per pricing and the algorithmic rate for (streams, K per stream, wgs)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi

M, N = int(os.environ.get("M", 1_000_000)), 252
def params(i): return _ffi.make_params(semantics="reference", n_paths=M, n_steps=N, seed=42, stream=i)

def run(nstreams, k, wgs, reps=2):
    ctxs = [_ffi.Context(0) for _ in range(nstreams)]
    for c in ctxs:
        c.set_option("seq_step_k", k); c.set_option("seq_step_wgs", wgs)
        c.price_american_seq([params(900 + i) for i in range(k)])
    def work(c, base):
        for r in range(reps):
            c.price_american_seq([params(base + r * k + i) for i in range(k)])
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, 1000 * i)) for i, c in enumerate(ctxs)]
    [t.start() for t in th]; [t.join() for t in th]
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    n = nstreams * k * reps
    gen = 0.167e-3  # generator per pricing at 1M paths (measured)
    sweep = dt / n - gen * M / 1e6
    print(f"streams {nstreams} K/stream {k} wgs/launch {wgs}: {1e3*dt/n:.3f} ms/pricing, sweep {1e6*sweep/N:.2f} us per pricing-step "
          f"-> {13.0*M/(sweep/N)/1e12:.2f} TB/s = {13.0*M/(sweep/N)/8e12:.3f} of peak; {M*N*n/dt:.3g} path-steps/s", flush=True)
    for c in ctxs: c.close()

for cfg in [(1, 4, 256), (2, 2, 128), (2, 2, 256), (4, 1, 64), (4, 1, 128), (1, 8, 256), (2, 4, 128), (2, 4, 256), (4, 2, 64), (4, 2, 128), (2, 8, 128), (4, 4, 64), (4, 4, 128)]:
    run(*cfg)
