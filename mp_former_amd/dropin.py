"""Install the native ops behind the reference's import names."""
import sys


def install():
    """Register ``MultiScaleDeformableAttention`` in sys.modules so the reference's
    ``import MultiScaleDeformableAttention as MSDA`` (ops/functions/ms_deform_attn_func.py:22)
    binds to the HIP kernels."""
    from . import MultiScaleDeformableAttention as mod
    sys.modules["MultiScaleDeformableAttention"] = mod
    return mod
