#!/usr/bin/env python3
"""Record the call signatures of the reference's entry points for this path -> tests/golden/signatures.json.

The reference's sources are only PARSED (ast), never imported or copied: what is stored is, per function or method,
the ordered parameter names and the literal defaults -- the data a drop-in has to reproduce.  tests/
test_signatures_cpu.py holds the surfaces of options_model_amd against it.
usage: python tools/capture_signatures.py [/root/reference]
"""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "signatures.json")

WANTED = {
    "Options_model.py": ["price_american_option", "compute_curve_for_S0"],
    "options_model_v1.5.py": ["OptionPricer.__init__", "OptionPricer.price_american_option",
                              "OptionPricer.compute_curve_for_S0", "compute_curve_worker"],
    "options_model_2.py": ["OptionPricer.__init__", "OptionPricer.price_american_option",
                           "OptionPricer.compute_curve_for_S0", "compute_curve_worker"],
    "options_model_3/options_model_3.py": [
        "welford_batch_update", "monte_carlo_price_streaming", "RNGManager.__init__", "RNGManager.get_child_rng",
        "RNGManager.get_child_seed", "BlackScholesGreeks.greeks", "BlackScholesGreeks.black_scholes_price",
        "AdvancedOptionPricer.__init__", "AdvancedOptionPricer.price_european_streaming",
        "AdvancedOptionPricer.price_american_enhanced_lsm", "AdvancedOptionPricer.price_american_with_control_variate",
        "AdvancedOptionPricer.price_american_option", "AdvancedOptionPricer.compute_curve_for_S0",
        "compute_curve_worker_enhanced"],
    "options_model_3/option_model_3_gpu.py": [
        "simulate_bs_paths_torch", "simulate_bs_paths_torch_bandwidth_optimized", "simulate_heston_paths_torch",
        "AdvancedOptionPricer.__init__", "AdvancedOptionPricer.price_european_gpu",
        "AdvancedOptionPricer.price_american_enhanced_lsm_gpu", "AdvancedOptionPricer.price_american_with_control_variate",
        "AdvancedOptionPricer.price_american_option", "AdvancedOptionPricer.compute_curve_for_S0",
        "compute_curve_worker_gpu", "compute_multiple_S0_gpu_batch"],
    "options_model_3/heston_calibration.py": ["HestonPricer.__init__", "HestonPricer.price_european_option",
                                              "HestonPricer.price_options_batch"],
}


def literal(node):
    try:
        return repr(ast.literal_eval(node))
    except Exception:  # noqa: BLE001  (a name or call: keep its source form)
        return ast.unparse(node)


def signature(fn: ast.FunctionDef):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(pos) - len(a.defaults)) + [literal(d) for d in a.defaults]
    params = [{"name": n, "default": d} for n, d in zip(pos, defaults)]
    if params and params[0]["name"] in ("self", "cls"):
        params = params[1:]
    for k, d in zip(a.kwonlyargs, a.kw_defaults):
        params.append({"name": k.arg, "default": None if d is None else literal(d), "kwonly": True})
    return params


def main():
    out = {}
    for rel, names in WANTED.items():
        path = os.path.join(REF, rel)
        # tolerant parse: the GPU file has an indentation slip inside a class body; the definitions
        # wanted here all precede or follow it and parse function by function
        src = open(path).read()
        try:
            tree = ast.parse(src)
        except SyntaxError:
            lines = src.splitlines()
            bad = [i for i, ln in enumerate(lines) if ln.startswith("    self.") and not ln.startswith("        ")]
            for i in bad:
                lines[i] = "    " + lines[i]
            tree = ast.parse("\n".join(lines))
        found = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                found[node.name] = signature(node)
            elif isinstance(node, ast.ClassDef):
                for m in node.body:
                    if isinstance(m, ast.FunctionDef):
                        found[f"{node.name}.{m.name}"] = signature(m)
        out[rel] = {}
        for n in names:
            if n not in found:
                raise SystemExit(f"{rel}: {n} not found")
            out[rel][n] = found[n]
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"wrote {OUT}: {sum(len(v) for v in out.values())} signatures")


if __name__ == "__main__":
    main()
