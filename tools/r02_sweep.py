#!/usr/bin/env python3
"""Round-2 kernel experiments on one MI355X: every variant runs in its own process (the knobs are
environment variables read once per process) and prints one JSON line.

  python tools/r02_sweep.py step     per-step sweep: workgroup size x graph on/off, C2 and C3 shard
  python tools/r02_sweep.py pass1    lsm_pass1_kernel: diagnostic builds (arithmetic / loads / no reduce), tiling knobs
  python tools/r02_sweep.py heston   heston_paths_kernel: vector width, schemes
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
from options_model_amd import _ffi
spec = json.loads(sys.argv[1])
ctx = _ffi.Context(0)
for k, v in spec.get("options", {}).items():
    ctx.set_option(k, v)
M, N = spec["M"], spec.get("N", 252)
kw = dict(model=spec.get("model", "gbm"), is_put=spec.get("is_put", True), semantics=spec["sem"], n_paths=M, n_steps=N,
          seed=42, heston_scheme=spec.get("scheme", "reference"))
reps = spec.get("reps", 20)
ctx.price_american_seq([_ffi.make_params(stream=900 + i, **kw) for i in range(3)])
ctx.sync()
t0 = time.perf_counter()
outs = ctx.price_american_seq([_ffi.make_params(stream=i, **kw) for i in range(reps)])
ctx.sync()
dt = (time.perf_counter() - t0) / reps
o = outs[0]
print(json.dumps(dict(spec=spec, ms_wall=1e3 * dt, ms_paths=o["ms_paths"], ms_seq_avg=o["ms_total"], ms_lsm_avg=o["ms_lsm"],
                      ms_pass1=o["ms_pass1"], ms_pass2=o["ms_pass2"], price=outs[-1]["price"])))
ctx.close()
''' % dict(root=ROOT)


def run(spec, env=None):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in (env or {}).items()})
    out = subprocess.run([sys.executable, "-c", WORKER, json.dumps(spec)], env=e, capture_output=True, text=True,
                         timeout=600)
    if out.returncode != 0:
        print(json.dumps(dict(spec=spec, env=env, error=out.stderr[-800:])), flush=True)
        return None
    d = json.loads(out.stdout.strip().splitlines()[-1])
    d["env"] = env or {}
    print(json.dumps(d), flush=True)
    return d


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "step"
    if what == "step":
        for M in (1_000_000, 8_000_000):
            for sem in ("reference", "textbook"):
                for blk in (1024, 512):
                    for graph in (1, 0):
                        run(dict(sem=sem, M=M, reps=10 if M > 2_000_000 else 20, options=dict(step_graph=graph)),
                            dict(OMC_STEP_BLOCK=blk))
    elif what == "pass1":
        for M in (1_000_000, 8_000_000):
            run(dict(sem="two_pass", M=M))
            for diag in (1, 2, 3):  # measurement kernels: only in a library built with -DOMC_DIAG_BUILD (the child
                # process rebuilds when the recorded flags differ; the default library is rebuilt by the next default run)
                run(dict(sem="two_pass", M=M), dict(OMC_PASS1_DIAG=diag, OMC_HIPCC_FLAGS="-DOMC_DIAG_BUILD"))
            for tpw in (2, 8):
                run(dict(sem="two_pass", M=M), dict(OMC_PASS1_TPW=tpw))
            for tch in (16, 64):
                run(dict(sem="two_pass", M=M), dict(OMC_PASS1_TCHUNK=tch))
    elif what == "heston":
        for scheme in ("reference", "full_truncation"):
            for vec in (4, 2, 1):
                run(dict(sem="two_pass", M=4_000_000, model="heston", is_put=False, scheme=scheme, reps=8,
                         options=dict(heston_vec=vec)))
        for vec in (4, 2):
            run(dict(sem="two_pass", M=4_000_000, model="gbm", reps=8, options=dict(gbm_vec=vec)))


if __name__ == "__main__":
    main()
