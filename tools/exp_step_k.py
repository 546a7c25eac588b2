"""Per-step reference flow: time per launch against the number of pricings sharing a launch (option seq_step_k).
usage: exp_step_k.py [paths] [model]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model = sys.argv[2] if len(sys.argv) > 2 else "gbm"
N = 252
ctx = _ffi.Context(0)
def params(i): return _ffi.make_params(model=model, is_put=(model == "gbm"), semantics="reference", n_paths=M, n_steps=N, seed=42, stream=i,
                                        heston_scheme="full_truncation" if model == "heston" else "reference")
for k in (1, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32):
    ctx.set_option("seq_step_k", k)
    ps = [params(900 + i) for i in range(max(k, 2))]
    ke = ctx.seq_step_width(ps)
    if ke != k and k > 1:
        print(f"K {k}: limited to {ke} by the byte budget"); 
        if ke < k: continue
    ctx.price_american_seq(ps)
    reps = 2 * k if k > 1 else 8
    t0 = time.perf_counter()
    outs = ctx.price_american_seq([params(i) for i in range(reps)])
    dt = time.perf_counter() - t0
    sweep = sum(o["ms_total"] for o in outs) / len(outs) - outs[0]["ms_paths"]
    print(f"M {M} {model} K {k:2d}: {1e3*sweep*k/N:7.2f} us per launch, {1e3*sweep/N:5.2f} us per pricing-step = {12.0*M/(sweep*1e-3/N)/8e12:.3f} of 8 TB/s; "
          f"pricing {1e3*dt/reps:.3f} ms = {M*N*reps/dt:.3g} path-steps/s", flush=True)
ctx.close()
