#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop from the gfx950 ISA (hipcc -save-temps), priced with the issue
intervals measured by tools/ubench.hip (profiles/r02c_ubench.txt).  usage: isa_mix.py <source.hip> <mangled-name-substring> <outputs per loop body>"""
import collections
import os
import re
import subprocess
import sys
import tempfile

# cycles per wave-instruction per SIMD at saturation (profiles/r02c_ubench.txt)
COST = collections.OrderedDict([
    ("v_mad_u64_u32", 6.1), ("v_mul_hi_u32", 5.4), ("v_mul_lo_u32", 5.4),
    ("v_exp_f32", 8.0), ("v_log_f32", 8.0), ("v_sqrt_f32", 8.0), ("v_sin_f32", 8.0), ("v_cos_f32", 8.0), ("v_rcp_f32", 8.0),
    ("v_pk_", 4.2), ("_f64", 4.3), ("v_cvt_f64", 4.9),
])


def cost(op):
    for k, v in COST.items():
        if k in op:
            return v
    return 2.1 if op.startswith("v_") else 0.0  # scalar / memory instructions issue beside the VALU


def main():
    src, sym, outputs = sys.argv[1], sys.argv[2], float(sys.argv[3])
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-save-temps",
                               os.path.abspath(src), "-o", "x.o"], cwd=d, stderr=subprocess.DEVNULL)
        s = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
    m = re.search(r"^(\S*" + re.escape(sym) + r"\S*):", s, re.M)
    body = s[m.start():s.index(".Lfunc_end", m.start())]
    # whole kernel body: the time loop is unrolled once per Philox block, prologue and epilogue are a few dozen
    # instructions (addresses, the t = 0 row) and are counted too -- an upper bound on the loop's mix
    loop = body
    c = collections.Counter()
    for ln in loop.split("\n"):
        ln = ln.strip()
        if ln and not ln.startswith((".", ";", "/")) and not ln.endswith(":"):
            c[ln.split()[0]] += 1
    tot = sum(c.values())
    cyc = sum(cost(k) * v for k, v in c.items())
    print(f"{m.group(1)}\nkernel body: {tot} instructions, {outputs:.0f} outputs per pass -> {tot / outputs:.1f} instructions, "
          f"{cyc / outputs:.1f} VALU issue cycles per output (per wave-instruction = 64 outputs)")
    for k, v in c.most_common():
        print(f"  {k:28s} {v:5d}  x {cost(k):4.1f} cyc = {cost(k) * v:7.1f}")


if __name__ == "__main__":
    main()
