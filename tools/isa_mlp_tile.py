#!/usr/bin/env python3
"""Static instruction budget of ONE 32-row tile of the fused trainer (mlp_train_kernel<L>): compiles csrc/omc_mlp.hip
with -save-temps, takes the kernel's largest backward-branch loop (the tile loop) and counts its instructions by pipe.
usage: isa_mlp_tile.py [layers 2|3]   (run in the build container: hipcc cross-compiles, no GPU needed)"""
import collections, os, re, subprocess, sys, tempfile

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-save-temps",
                           os.path.join(root, "options_model_amd", "csrc", "omc_mlp.hip"), "-o", "x.o"], cwd=d,
                          stderr=subprocess.DEVNULL)
    s = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
m = re.search(r"^(\S*mlp_train_kernelILi%dEE\S*):" % L, s, re.M)
lines = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
labels = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(lines):
    mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((i - labels[mm.group(1)], labels[mm.group(1)], i))
_, st, en = max(loops)
c = collections.Counter()
for l in lines[st:en + 1]:
    l = l.strip()
    if l and not l.startswith((".", ";", "/")) and not l.endswith(":"):
        c[l.split()[0]] += 1
cat = collections.Counter()
for k, v in c.items():
    if k.startswith("v_mfma"): cat[k] += v
    elif k.startswith("v_accvgpr"): cat["v_accvgpr_* (MFMA result moves)"] += v
    elif k.startswith("v_"): cat["other VALU"] += v
    elif k.startswith("ds_"): cat["LDS " + ("read" if "read" in k else "write")] += v
    elif k.startswith("s_waitcnt"): cat["s_waitcnt"] += v
    elif k.startswith("s_nop"): cat["s_nop"] += v
    elif k.startswith("s_"): cat["SALU"] += v
    else: cat["vector memory"] += v
print(f"mlp_train_kernel<{L}>: tile loop = {sum(c.values())} instructions (both branches of the dropout test are in the count:")
print("  the no-dropout branch is 64 v_max_f32 per hidden layer, the dropout branch ~4.5 VALU per activation)")
for k, v in cat.most_common():
    print(f"  {k:36s}{v:6d}")
mf = 64 * c.get("v_mfma_f32_32x32x2_f32", 0) + 32 * c.get("v_mfma_f32_16x16x4_f32", 0)
print(f"MFMA pipe: {c.get('v_mfma_f32_32x32x2_f32', 0)} x 64 + {c.get('v_mfma_f32_16x16x4_f32', 0)} x 32 cycles = {mf} cycles per tile "
      f"(16 / 8 passes of 4 cycles)")
print("top VALU opcodes:", ", ".join(f"{k} {v}" for k, v in c.most_common(14) if k.startswith("v_") and not k.startswith("v_mfma")))
