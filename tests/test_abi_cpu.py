"""CPU suite, part 2: the C-ABI library loads without a GPU and exports every symbol that
include/omc.h declares; argument checks that need no device work."""
import ctypes as C
import os
import re

import pytest

from options_model_amd import _build, _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "omc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(omc_[a-z0-9_]+)\s*\(", txt)) - {"omc_allreduce_fn"})


def test_library_is_built_in_tree():
    lib = _ffi.load_library()
    assert os.path.samefile(os.path.dirname(_build.LIB), os.path.join(ROOT, "options_model_amd", "lib"))
    assert lib.omc_abi_version() == _ffi.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    lib = _ffi.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/omc.h but not exported by libomc.so"
        assert s in _ffi.SIGNATURES, f"{s} has no ctypes signature in _ffi.py"
    assert sorted(_ffi.SIGNATURES) == syms


def test_struct_layouts_match_header():
    assert C.sizeof(_ffi.Params) == 6 * 4 + 8 + 10 * 8 + 3 * 8
    assert C.sizeof(_ffi.Result) == 5 * 8 + 4 * 8 + 5 * 8 + 8 + 8


def test_code_object_targets_gfx950():
    blob = open(_build.LIB, "rb").read()
    assert b"gfx950" in blob


def test_no_device_fails_loudly_not_silently():
    if _ffi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises((_ffi.OmcError, ValueError)):
        _ffi.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "options_model_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "omc_oracle" not in src and "libomc_oracle" not in src, f


def test_c_host_example_compiles_against_the_header_and_library(tmp_path):
    """examples/price_american.c is the binding a non-Python host would write: it must compile with
    a plain C compiler against include/omc.h and link against libomc.so (no GPU needed for that)."""
    import shutil
    import subprocess
    from options_model_amd import _build
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = _build.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "price_american"
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
           os.path.join(root, "examples", "price_american.c"), "-o", str(exe), "-L", os.path.dirname(lib), "-lomc",
           "-lm", "-Wl,-rpath," + os.path.dirname(lib)]
    subprocess.run(cmd, check=True)
    assert exe.exists()


def test_facade_n_gpus_never_prices_on_fewer_gpus(monkeypatch):
    """price_american_option(..., n_gpus=N) shards over N ranks.  From a plain process it starts the
    ranks itself (launcher.RankPool); here there is no GPU, so every rank fails to bring its context up and the call
    raises with the ranks named -- never a silent single-GPU pricing, and no GPU call in this process.  Inside a job
    of another size it refuses outright."""
    import time

    import pytest

    from options_model_amd import launcher, price_american_option
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    t0 = time.monotonic()
    with pytest.raises(launcher.RankError, match=r"rank \d"):  # whichever rank reports first
        price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 10_000, 50, n_gpus=2)
    assert time.monotonic() - t0 < 120
    launcher.close_pools()
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="inside a 4-rank job"):
        price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 10_000, 50, n_gpus=2)
    with pytest.raises(ValueError, match="n_gpus"):
        price_american_option(100.0, 100.0, 0.05, 0.2, 1.0, 10_000, 50, n_gpus=0)
