"""Drop-in for the reference's compiled extension module ``MultiScaleDeformableAttention``
(pybind: mask2former/modeling/pixel_decoder/ops/src/vision.cpp:18-21), so that the reference's
``ops/functions/ms_deform_attn_func.py:21-29,36,46`` works UNCHANGED:

    import mp_former_amd.dropin; mp_former_amd.dropin.install()
    import MultiScaleDeformableAttention as MSDA      # -> this module

or put ``mp_former_amd/`` itself on PYTHONPATH.  See INTEGRATION.md.
"""
from mp_former_amd.msda import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
