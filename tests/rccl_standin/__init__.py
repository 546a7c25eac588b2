"""Test infrastructure: build / locate the shared-memory stand-in for librccl (rccl_standin.cpp).

Only tests (and __graft_entry__.build(), which compiles it so that it travels to the GPU box) use this;
the product never does -- libomc.so opens whatever OMC_RCCL_LIB names, and nothing in options_model_amd/
sets that variable."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "rccl_standin.cpp")
OUT_DIR = os.path.join(_HERE, "_build")
LIB = os.path.join(OUT_DIR, "librccl_standin.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # host code only (no kernels): hipcc for the HIP / RCCL include paths and the runtime library
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB, SRC, "-lrt", "-lpthread"])
    return LIB
