"""Drop-in conformance of the Python surfaces: every entry point of the reference for this path (tests/golden/
signatures.json, parsed from the reference's sources by tools/capture_signatures.py) has a counterpart here that
takes the same parameters in the same order with the same defaults.  Extra parameters are allowed only after the
reference's own ones and only with defaults (so that every call the reference accepts means the same thing here).
No GPU, no HIP call: importing the surfaces must not touch the device."""
import ast
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIGS = json.load(open(os.path.join(HERE, "golden", "signatures.json")))

SURFACE = {
    "Options_model.py": "options_model_amd.compat.Options_model",
    "options_model_v1.5.py": "options_model_amd.compat.options_model_v1_5",
    "options_model_2.py": "options_model_amd.compat.options_model_2",
    "options_model_3/options_model_3.py": "options_model_amd.pricer",
    "options_model_3/option_model_3_gpu.py": "options_model_amd.compat.option_model_3_gpu",
    "options_model_3/heston_calibration.py": "options_model_amd.heston_pricer",
}
CASES = [(f, n) for f, v in sorted(SIGS.items()) for n in sorted(v)]


def _resolve(module, dotted):
    obj = importlib.import_module(module)
    for part in dotted.split("."):
        obj = getattr(obj, part)
    return obj


def _same_default(ref_repr, value):
    try:
        return ast.literal_eval(ref_repr) == value
    except (ValueError, SyntaxError):
        return False


@pytest.mark.parametrize("ref_file,name", CASES)
def test_surface_accepts_the_references_call_signature(ref_file, name):
    fn = _resolve(SURFACE[ref_file], name)
    params = [p for p in inspect.signature(fn).parameters.values() if p.name not in ("self", "cls")]
    ref = SIGS[ref_file][name]
    ref_pos = [p for p in ref if not p.get("kwonly")]
    assert len(params) >= len(ref_pos), f"{name}: fewer parameters than the reference"
    for i, rp in enumerate(ref_pos):
        p = params[i]
        assert p.name == rp["name"], f"{name}: parameter {i} is {p.name!r}, the reference has {rp['name']!r}"
        assert p.kind in (p.POSITIONAL_OR_KEYWORD, p.POSITIONAL_ONLY, p.VAR_POSITIONAL)
        if rp["default"] is None:
            pass  # required in the reference; a default here only makes the surface more permissive
        else:
            assert p.default is not p.empty and _same_default(rp["default"], p.default), \
                f"{name}.{p.name}: default {p.default!r}, the reference has {rp['default']}"
    for p in params[len(ref_pos):]:  # ours only: must not change the meaning of any reference call
        assert p.kind in (p.KEYWORD_ONLY, p.VAR_KEYWORD, p.VAR_POSITIONAL) or p.default is not p.empty, \
            f"{name}: extra parameter {p.name!r} without a default"


def test_importing_the_surfaces_makes_no_device_call():
    """The reference's UIs spawn worker processes that import these modules: import must stay cheap and must not
    initialise HIP (a context is created on first use)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import options_model_amd, options_model_amd.compat.Options_model, options_model_amd.compat.options_model_2\n"
            "import options_model_amd.compat.options_model_v1_5, options_model_amd.compat.option_model_3_gpu\n"
            "import options_model_amd.heston_pricer\n"
            "from options_model_amd import _ffi\n"
            "assert not _ffi._default_ctx, 'a context was created at import'\n"
            "assert 'torch' not in sys.modules, 'torch was imported by the surfaces'\n"
            "print('ok')\n") % os.path.dirname(HERE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
