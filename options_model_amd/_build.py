"""In-tree build of libomc.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
options_model_amd/lib/libomc.so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libomc.so")
SOURCES = ["omc_paths.hip", "omc_lsm.hip", "omc_batch.hip", "omc_mlp.hip", "omc_contnet.hip", "omc_rows.hip", "omc_ols7.hip", "omc_comm.hip", "omc_p2p.hip",
           "omc_api.hip"]
HEADERS = ["omc_device.h", "omc_kernels.h", "omc_lsm_dev.h", "omc_paths_dev.h", "omc_contnet_dev.h", "omc_batch.h", "omc_comm.h", "omc_p2p.h",
           os.path.join("..", "..", "include", "omc.h")]
ARCH = "gfx950"
# The extra compiler flags the library was built with (OMC_HIPCC_FLAGS: experiment builds such as
# -DOMC_DIAG_BUILD or -DOMC_P2_U=16) are recorded next to it: a library left behind by an experiment is
# stale for every process that does not ask for the same flags, so tests / bench / profiles can never
# silently run a non-default build.
FLAGS_STAMP = os.path.join(LIBDIR, "libomc.flags")


def _flags() -> str:
    return " ".join(os.environ.get("OMC_HIPCC_FLAGS", "").split())


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libomc.so cannot be built (ROCm toolchain required)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    try:
        built_with = open(FLAGS_STAMP).read().strip()
    except OSError:
        built_with = ""
    if built_with != _flags():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c",
               "-Wall", "-Wno-unused-function", os.path.join(CSRC, s), "-o", o]
        cmd[1:1] = _flags().split()  # experiments (-D..., -save-temps)
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    for f in os.listdir(LIBDIR):  # objects of sources that no longer exist must not travel to the GPU box
        if f.endswith(".o") and os.path.join(LIBDIR, f) not in objs:
            os.remove(os.path.join(LIBDIR, f))
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    with open(FLAGS_STAMP, "w") as f:
        f.write(_flags() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
