#!/usr/bin/env python3
"""Which shader clock does the chip hold while the two-pass LSM kernels run back to back?  (OMC_PASS1_DIAG=1: the
arithmetic-only build of pass 1.)  Samples rocm-smi while a thread keeps the GPU busy."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from options_model_amd import _ffi

M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 252
ctx = _ffi.Context(0)
S = ctx.gbm_paths(M, N, 100.0, 0.05, 0.2, 1.0, seed=42)
stop = False
ms = []
def work():
    while not stop:
        ms.append(ctx.lsm_poly(S, 100.0, 0.05, 1.0, True, "two_pass")["ms_pass1"])
t = threading.Thread(target=work); t.start()
time.sleep(1.0)
samples = []
for _ in range(8):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pwr = re.findall(r"Power \(W\): ([0-9.]+)", out)
    samples.append((sclk[:1], pwr[:1]))
    time.sleep(0.2)
stop = True; t.join()
import statistics
print("diag", os.environ.get("OMC_PASS1_DIAG", "-"), "pass1 ms median", round(statistics.median(ms[len(ms)//2:]), 4), "samples", samples)
