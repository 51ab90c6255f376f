"""ORACLE — test infrastructure only (never imported by mp_former_amd).

numpy front-end of the plain-C MSDA restatement in ``msda_ref.c`` (forward and backward of
multi-scale deformable attention; reference:
mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-408 and
ops/functions/ms_deform_attn_func.py:52-72 for the python path it is pinned against).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_msda.so")
_lib = None


def build(force=False):
    """Compile msda_ref.c with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "msda_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_msda.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, level_start, loc, attn):
    dt = value.dtype
    assert dt in (np.float32, np.float64), dt
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    level_start = np.ascontiguousarray(level_start, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and level_start.shape == (L,)
    assert attn.shape == (N, Lq, M, L, P)
    # every level's rows lie inside value (the reference indexes value by level_start_index only — ms_deform_im2col_cuda.cuh:268 —
    # so levels need not be stored back to back)
    assert (level_start >= 0).all() and (level_start + shapes[:, 0] * shapes[:, 1] <= S).all()
    return value, shapes, level_start, loc, attn, (N, S, M, D, L, Lq, P)


def msda_forward(value, shapes, level_start, loc, attn):
    """-> out [N, Lq, M*D]"""
    value, shapes, level_start, loc, attn, dims = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_load(), "oracle_msda_forward_" + ("f32" if value.dtype == np.float32 else "f64"))
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(attn), _ptr(out),
       *[ctypes.c_int(v) for v in dims])
    return out


def msda_backward(value, shapes, level_start, loc, attn, grad_out):
    """-> (grad_value, grad_loc, grad_attn)"""
    value, shapes, level_start, loc, attn, dims = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    assert grad_out.shape == (N, Lq, M * D)
    gv = np.zeros_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_load(), "oracle_msda_backward_" + ("f32" if value.dtype == np.float32 else "f64"))
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(attn), _ptr(grad_out),
       _ptr(gv), _ptr(gl), _ptr(ga), *[ctypes.c_int(v) for v in dims])
    return gv, gl, ga
