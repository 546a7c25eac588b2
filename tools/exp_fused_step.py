#!/usr/bin/env python3
"""EXPERIMENT: the optimizer step of the reference's default call without its Adam launch.

OMC_MLP_FUSED = 0  the product: mlp_train_q16_kernel + mlp_adam_kernel, two dependent launches per step
              = 1  variant (a): ONE launch per step; the tile workgroups publish their partials, meet at a barrier of the
                   step's tile workgroups and apply Adam to a slice of the parameters each (reduce-scatter through L2)
              = 2  variant (b): ONE launch per epoch; a second barrier per step replaces the launch boundary
(the variable is read once per process, so every variant runs in a child process).  Every variant trains the same network
on the same rows with the same seeds; the parent compares parameters, Adam moments and losses BIT FOR BIT with variant 0
and prints the time per optimizer step.

usage: exp_fused_step.py [rows] [batch] [epochs] [hidden] [layers] [dropout]     (defaults: config 1's 225,057 rows, 256, 4, 128, 3, 0.1)

NOTE: both variants LOST (39 against 22.3 us per step, profiles/r06_fused_step_experiment.txt) and their kernels
(mlp_train_q16_fused_kernel, the OMC_MLP_FUSED switch) were taken out of the product again: to re-run this script check out
commit 38a8e3a, where they live.  On the current tree every variant runs the product's two launches.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(rows, batch, epochs, hidden, layers, drop):
    sys.path.insert(0, ROOT)
    import torch
    from options_model_amd import _ffi, nn_regressor as nnr
    dev = torch.device("cuda", 0)
    ctx = _ffi.default_context(0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    data = torch.randn(rows, 8, device=dev, generator=g)
    torch.manual_seed(99)
    net = nnr.make_net(7, hidden, layers, 0.1).to(dev)
    p = nnr.flatten_params(net)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    torch.cuda.synchronize()
    args = (p.data_ptr(), m.data_ptr(), v.data_ptr())
    step, losses = 0, []
    loss, step = ctx.mlp_train_epoch(data.data_ptr(), min(rows, 8 * batch + 37), batch, *args, step, 1e-3, drop, 5, shuffle_key=3,
                                     layers=layers, hidden=hidden)  # warm (also a ragged last minibatch)
    losses.append(loss)
    t0 = time.perf_counter()
    for e in range(epochs):
        loss, step = ctx.mlp_train_epoch(data.data_ptr(), rows, batch, *args, step, 1e-3, drop, 5, shuffle_key=8 + e, layers=layers,
                                         hidden=hidden)
        losses.append(loss)
    dt = time.perf_counter() - t0
    nsteps = epochs * ((rows + batch - 1) // batch)
    h = hashlib.sha256()
    for t in (p, m, v):
        h.update(t.cpu().numpy().tobytes())
    print("RESULT " + json.dumps(dict(fused=int(os.environ.get("OMC_MLP_FUSED", "0")), us_per_step=1e6 * dt / nsteps, steps=nsteps,
                                      losses=losses, sha256=h.hexdigest(), finite=bool(torch.isfinite(p).all()))), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        a = sys.argv[2:]
        child(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]))
        return
    a = sys.argv[1:]
    rows, batch, epochs = int(a[0]) if a else 225_057, int(a[1]) if len(a) > 1 else 256, int(a[2]) if len(a) > 2 else 4
    hidden, layers, drop = int(a[3]) if len(a) > 3 else 128, int(a[4]) if len(a) > 4 else 3, float(a[5]) if len(a) > 5 else 0.1
    out = {}
    for fused in (0, 1, 2, 0):
        env = dict(os.environ, OMC_MLP_FUSED=str(fused))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(rows), str(batch), str(epochs), str(hidden),
                            str(layers), str(drop)], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if p.returncode != 0 or not line:
            print(f"variant {fused}: FAILED rc={p.returncode}: {p.stderr[-600:]}")
            continue
        r = json.loads(line[0][7:])
        key = f"{fused}" if f"{fused}" not in out else f"{fused} (again)"
        out[key] = r
        base = out.get("0")
        same = base is not None and r["sha256"] == base["sha256"] and r["losses"] == base["losses"]
        print(f"rows {rows} batch {batch} {layers} x {hidden} dropout {drop}: OMC_MLP_FUSED={fused}: {r['us_per_step']:.2f} us per optimizer "
              f"step over {r['steps']} steps; final loss {r['losses'][-1]:.9f}; parameters / moments / losses "
              f"{'BIT-EQUAL to' if same else 'DIFFERENT from'} variant 0", flush=True)


if __name__ == "__main__":
    main()
