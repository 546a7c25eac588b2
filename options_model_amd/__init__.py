"""options_model_amd -- MI355X-native hot path of Levicoz/Options-model.

American-option Monte-Carlo pricing (GBM / Heston path generation -> Longstaff-Schwartz
backward induction -> discounted mean) as hand-written HIP kernels for gfx950 behind a C ABI
(include/omc.h, lib/libomc.so), with the reference's Python call surface kept as a drop-in:

    from options_model_amd import price_american_option           # north-star facade
    from options_model_amd import AdvancedOptionPricer, RNGManager   # options_model_3.py surface
    from options_model_amd.compat import Options_model, options_model_2   # v1 / v2 surfaces

Nothing here touches the GPU at import time.
"""
from ._ffi import OmcError, Context, default_context, device_count, load_library  # noqa: F401

__all__ = ["OmcError", "Context", "default_context", "device_count", "load_library"]


def __getattr__(name):
    # lazy: keep `import options_model_amd` cheap and free of side effects
    if name in ("price_american_option", "PriceResult", "price_european_option"):
        from . import api
        return getattr(api, name)
    if name in ("AdvancedOptionPricer", "RNGManager", "BlackScholesGreeks", "welford_batch_update",
                "monte_carlo_price_streaming", "compute_curve_worker_enhanced"):
        from . import pricer
        return getattr(pricer, name)
    raise AttributeError(name)
