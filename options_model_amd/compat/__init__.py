"""Call-surface mirrors of the reference's older generations, so its Streamlit UIs and scripts
can switch imports without touching call sites:

    options_ui.py:13          from Options_model import price_american_option, compute_curve_for_S0
    options_model_2_ui.py:14  from options_model_2 import compute_curve_worker
becomes
    from options_model_amd.compat.Options_model import price_american_option, compute_curve_for_S0
    from options_model_amd.compat.options_model_2 import compute_curve_worker
"""
