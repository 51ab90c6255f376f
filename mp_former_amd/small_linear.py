"""bf16 Linear for the decoder's query-side activations (a few hundred rows) on the small-row MFMA GEMM
(csrc/small_gemm.hip; include/mpformer_hip.h mpf_small_gemm_bf16).

``small_linear(x, w, b, relu)`` == ``relu?(F.linear(x, w, b))`` for bf16 CUDA tensors (what autocast
makes of the nn.Linear / nn.MultiheadAttention projections of mask2former_transformer_decoder.py:19-206),
with the ReLU, its backward gate and the bias gradient folded into the three GEMMs of the layer.
Same rounding points as the library path: fp32 accumulation, bf16 results.  GPU only.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

MAX_ROWS = 1024     # above this the library GEMMs have enough tiles to fill the chip


def _stream(t):
    return _lib.stream_ptr(t.device)


def small_gemm(a, a_rs, a_ks, b, b_rs, b_ks, I, J, Kc, bias=None, gate=None, relu=False, rowsum=False, cin=None):
    """C[I, J] = sum_k A(i,k) B(j,k) (+bias) (ReLU) with explicit element strides; returns (C, rowsum_a or None)."""
    c = torch.empty((I, J), dtype=torch.bfloat16, device=a.device)
    rs = torch.empty((I,), dtype=torch.bfloat16, device=a.device) if rowsum else None
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_small_gemm_bf16(
            a.data_ptr(), a_rs, a_ks, gate.data_ptr() if gate is not None else None, b.data_ptr(), b_rs, b_ks,
            bias.data_ptr() if bias is not None else None, cin.data_ptr() if cin is not None else None,
            cin.stride(0) if cin is not None else 0, c.data_ptr(), J, rs.data_ptr() if rowsum else None,
            I, J, Kc, 1 if relu else 0, _stream(a))
    _lib.check(code, "mpf_small_gemm_bf16")
    return c, rs


class MpfSmallGemmItem(ctypes.Structure):
    """include/mpformer_hip.h MpfSmallGemmItem."""
    _fields_ = [("a", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p),
                ("rowsum_a", ctypes.c_void_p), ("a_rs", ctypes.c_int64), ("a_ks", ctypes.c_int64), ("a_bs", ctypes.c_int64),
                ("b_rs", ctypes.c_int64), ("b_ks", ctypes.c_int64), ("ldc", ctypes.c_int64), ("a_blk", ctypes.c_int),
                ("I", ctypes.c_int), ("J", ctypes.c_int), ("Kc", ctypes.c_int)]


def weight_grads_grouped(problems):
    """[(dy [R, J], x [R, K], gate or None)] -> [(dW [J, K], db [J])]: up to 8 weight-gradient problems
    dW = (dy gated)^T . x in ONE launch (mpf_small_gemm_bf16_group)."""
    items = (MpfSmallGemmItem * len(problems))()
    outs = []
    for it, (dy, x, gate) in zip(items, problems):
        R, J = dy.shape
        K = x.shape[1]
        dw = torch.empty((J, K), dtype=torch.bfloat16, device=dy.device)
        db = torch.empty((J,), dtype=torch.bfloat16, device=dy.device)
        it.a, it.gate, it.b, it.c, it.rowsum_a = dy.data_ptr(), (gate.data_ptr() if gate is not None else None), x.data_ptr(), \
            dw.data_ptr(), db.data_ptr()
        it.a_rs, it.a_ks, it.a_bs, it.b_rs, it.b_ks, it.ldc = 1, dy.stride(0), 0, 1, x.stride(0), K
        it.a_blk, it.I, it.J, it.Kc = 0, J, K, R
        outs.append((dw, db))
    dev = problems[0][0].device
    with _lib.device_guard(dev):
        code = _lib.lib().mpf_small_gemm_bf16_group(items, len(problems), _stream(problems[0][0]))
    _lib.check(code, "mpf_small_gemm_bf16_group")
    return outs


def usable(x, w, b=None):
    """Shapes / dtypes the kernel takes (callers use F.linear otherwise)."""
    K = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and (b is None or b.dtype == torch.bfloat16)
            and x.numel() // max(K, 1) <= MAX_ROWS and K % 8 == 0 and w.shape[0] % 8 == 0 and x.numel() > 0)


class _SmallLinear(Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        w = w.contiguous()
        M, N = x2.shape[0], w.shape[0]
        y, _ = small_gemm(x2, x2.stride(0), 1, w, K, 1, M, N, K, bias=b, relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.has_bias = b is not None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        g2 = gy.reshape(M, N)
        if g2.stride(1) != 1 or g2.stride(0) % 8 or g2.data_ptr() % 16 or g2.dtype != torch.bfloat16:
            g2 = g2.to(torch.bfloat16).contiguous()
        if y is not None and g2.stride(0) != N:
            g2 = g2.contiguous()        # the gate is addressed like the gradient
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[m, k] = sum_n dY[m, n] W[n, k]: A = dY (contraction-contiguous), B(k, n) = W[n, k] (row-contiguous)
            dx, _ = small_gemm(g2, g2.stride(0), 1, w, 1, K, M, K, N, gate=y)
            dx = dx.view(*gy.shape[:-1], K)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # dW[n, k] = sum_m dY[m, n] x[m, k]; db[n] = sum_m dY[m, n] (row sums of the A operand)
            dw, db = small_gemm(g2, 1, g2.stride(0), x2, 1, x2.stride(0), N, K, M, gate=y, rowsum=ctx.has_bias)
        return dx, dw, db, None


def small_linear(x, w, b=None, relu=False):
    if not x.is_cuda:
        raise RuntimeError("mp_former_amd small_linear runs on the GPU only (no CPU fallback)")
    return _SmallLinear.apply(x, w, b, relu)


# ---------------------------------------------------------------------------------------------------
# bf16 Linear with MANY rows (the key / value projections of the cross-attention, the batched prediction heads): forward
# and input gradient on the native tall GEMM (mpf_tall_gemm_bf16; the library only for shapes it does not take); the weight
# gradient dW = dY^T x — a 256x256 output contracted over thousands of rows, 57-66 us in the library plus a separate bias
# reduction — on the split-over-rows MFMA kernel of gemm3.hip in its one-product bf16 form (mpf_gemm_nt_bf16), bias gradient
# included.
_nt_ws = {}


def _nt_workspace(device, nbytes):
    w = _nt_ws.get(device)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _nt_ws[device] = w
    return w


def gemm_nt_bf16(a, b, want_csum=True):
    """a [R, M], b [R, N] bf16 (unit column stride) -> (a^T b [M, N] bf16, column sums of a [M] bf16 or None)."""
    from .gemm3 import pick_rows_per_split
    R, M = a.shape
    N = b.shape[1]
    rps = pick_rows_per_split(R, ((M + 127) // 128) * ((N + 127) // 128))
    lib = _lib.lib()
    ws = _nt_workspace(a.device, lib.mpf_gemm_nt_bf16_workspace_bytes(R, M, N, rps))
    c = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    cs = torch.empty((M,), dtype=torch.bfloat16, device=a.device) if want_csum else None
    with _lib.device_guard(a.device):
        code = lib.mpf_gemm_nt_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(),
                                    cs.data_ptr() if want_csum else None, R, M, N, rps, ws.data_ptr(), ws.numel(), _stream(a))
    _lib.check(code, "mpf_gemm_nt_bf16")
    return c, cs


def tall_usable(x, w, b=None):
    K = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and (b is None or b.dtype == torch.bfloat16)
            and x.numel() // max(K, 1) > MAX_ROWS and K % 4 == 0 and w.shape[0] % 4 == 0)


def tall_gemm(a, b, bias=None):
    """a [M, K] . b [N, K]^T (+ bias [N]) -> [M, N], all bf16 (unit column strides), on mpf_tall_gemm_bf16
    (csrc/small_gemm.hip: operands straight from memory as MFMA fragments)."""
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    with _lib.device_guard(a.device):
        code = _lib.lib().mpf_tall_gemm_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                             bias.data_ptr() if bias is not None else None, c.data_ptr(), N, M, N, K, _stream(a))
    _lib.check(code, "mpf_tall_gemm_bf16")
    return c


def _tall_native_ok(a, b):
    """operand pair the native tall GEMM takes as it lies in memory: contraction-contiguous, 16-byte aligned rows, K % 32 == 0"""
    return (a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and a.shape[1] % 32 == 0 and a.stride(0) % 8 == 0
            and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and b.shape[0] % 4 == 0)


class _TallLinear(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        x2 = x.reshape(-1, x.shape[-1])
        if _tall_native_ok(x2, w) and (b is None or b.is_contiguous()):
            return tall_gemm(x2, w, b).view(*x.shape[:-1], w.shape[0])
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        K, N = x.shape[-1], w.shape[0]
        g2 = gy.reshape(-1, N)
        if g2.stride(1) != 1 or g2.dtype != torch.bfloat16:
            g2 = g2.to(torch.bfloat16).contiguous()
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX[m, k] = sum_n dY[m, n] W[n, k]: the native kernel contracts along contiguous memory, so it gets W^T (a
            # <= 400 KB copy; the weights of a level's three layers side by side)
            wt = w.t().contiguous() if N % 32 == 0 else None
            if wt is not None and _tall_native_ok(g2, wt):
                dx = tall_gemm(g2, wt).view(x.shape)
            else:
                dx = torch.matmul(g2, w).view(x.shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = gemm_nt_bf16(g2, x2, want_csum=ctx.has_bias)
        return dx, dw, db


def tall_linear(x, w, b=None):
    return _TallLinear.apply(x, w, b)
