"""mp_former_amd — MI355X-native (gfx950) hot path of IDEA-Research/MP-Former.

The product path is hand-written HIP behind a C ABI (``include/mpformer_hip.h`` ->
``libmpformer_hip.so``); this package is the thin host-side mirror of the reference's
operator / module interface for that path.  There is NO CPU fallback: every op raises if the
native library is missing or if it is handed CPU tensors (the reference's op does the same,
``ops/src/ms_deform_attn.h:43``).
"""
from . import _lib  # noqa: F401
from ._miopen import use_shipped_find_db  # noqa: F401  (an explicit call: importing the package changes no environment variable)
from .msda import (MSDeformAttn, MSDeformAttnFunction, ms_deform_attn_backward,  # noqa: F401
                   ms_deform_attn_forward)

__all__ = ["MSDeformAttn", "MSDeformAttnFunction", "ms_deform_attn_forward", "ms_deform_attn_backward"]
