#!/usr/bin/env python3
"""Timeline of ONE optimizer step of the reference's default shape (3 x 128, minibatch 256) on the 100 MHz wall clock:
needs an experiment build (OMC_HIPCC_FLAGS=-DOMC_Q16_EXP=64).  Also: the host's cost of enqueuing a step."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from options_model_amd import _ffi, nn_regressor as nnr

rows, batch, hidden, layers = 225057 * 3, 256, 128, 3
dev = torch.device("cuda", 0)
ctx = _ffi.default_context(0)
data = torch.randn(rows, 8, device=dev)
net = nnr.make_net(7, hidden, layers, 0.1).to(dev)
p = nnr.flatten_params(net); m = torch.zeros_like(p); v = torch.zeros_like(p)
torch.cuda.synchronize()
t0 = time.perf_counter()
loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, batch, p.data_ptr(), m.data_ptr(), v.data_ptr(), 0, 1e-3, 0.1, 1, shuffle_key=7,
                                 layers=layers, hidden=hidden)
dt = time.perf_counter() - t0
print(f"{step} steps, {1e6 * dt / step:.2f} us per step")
out = (C.c_ulonglong * 32)()
lib = ctx.lib
rc = lib.omc_debug_q16_stamps(out)
assert rc == 0, rc
s = list(out)
names = {0: "train entry", 1: "layer 0 done (inputs, W1 arrived)", 2: "layer 1 done", 3: "layer 2 done", 5: "output / dout / gwo done",
         6: "gW_2 done", 7: "dH_1 done", 8: "gW_1 done", 9: "dH_0 done", 12: "train exit (stores acknowledged)",
         16: "adam block 0 entry", 17: "adam block 0 exit", 18: "adam last block entry", 19: "adam last block exit",
         24: "next train entry"}
base = s[0]
for k in sorted(names):
    if s[k]:
        print(f"  {names[k]:40s} {(s[k] - base) * 0.01:8.2f} us")
