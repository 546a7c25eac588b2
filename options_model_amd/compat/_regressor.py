"""Which regressor the v1 / v2 drop-ins use at every time step.

"nn"   (default) the reference's own: a fresh ContNet per step (omc_contnet.hip, include/omc.h
       omc_price_american_contnet) -- prices land where the reference's land.
"poly" OLS on [1, u, u^2] per step (BASELINE.json's "polynomial LSM"), batched launches for curves.
Callers that cannot pass the keyword (the Streamlit UIs go through the curve workers) select with the
environment variable OMC_REGRESSOR, the same switch AdvancedOptionPricer reads.
"""
from __future__ import annotations

import os


def resolve(regressor=None) -> str:
    choice = regressor if regressor is not None else os.environ.get("OMC_REGRESSOR", "nn")
    choice = str(choice).lower()
    if choice not in ("nn", "poly"):
        raise ValueError("regressor must be 'nn' or 'poly'.")
    return choice
