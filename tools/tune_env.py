#!/usr/bin/env python3
"""Tuning sweep over environment-variable knobs (one subprocess per setting, since the library
reads them once).  usage: tune_env.py VAR=v1,v2,.. [VAR2=...]  -> ms per phase for C2 two_pass"""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNIP = r'''
import sys, json; sys.path.insert(0, %r)
from options_model_amd import _ffi
c = _ffi.Context(0)
sem = %r
p = lambda i: _ffi.make_params(semantics=sem, n_paths=%d, n_steps=252, seed=42, stream=i)
for i in range(3): c.price_american(p(100+i))
a = [c.price_american(p(i)) for i in range(15)]
print(json.dumps(dict(ms_paths=sum(x["ms_paths"] for x in a)/len(a), ms_lsm=sum(x["ms_lsm"] for x in a)/len(a), price=a[0]["price"])))
'''
def main():
    sem = os.environ.get("TUNE_SEM", "two_pass"); M = int(os.environ.get("TUNE_PATHS", "1000000"))
    axes = [(kv.split("=")[0], kv.split("=")[1].split(",")) for kv in sys.argv[1:]]
    for combo in itertools.product(*[v for _, v in axes]):
        env = dict(os.environ); env.update({k: v for (k, _), v in zip(axes, combo)})
        out = subprocess.run([sys.executable, "-c", SNIP % (ROOT, sem, M)], env=env, capture_output=True, text=True)
        print(dict(zip([k for k, _ in axes], combo)), out.stdout.strip() or out.stderr[-300:], flush=True)
if __name__ == "__main__":
    main()
