"""nn.Linear for fp32 activations with many rows on the native fp32 GEMM (csrc/gemm3.hip).

The per-layer route of the pixel decoder's encoder — ``MSDeformAttn.forward`` (ops/modules/ms_deform_attn.py:95-124) and the
FFN of ``MSDeformAttnTransformerEncoderLayer`` (msdeformattn.py:116-131) when they are called as modules instead of through
the fused encoder node — sees [N * S, C] activations with N * S = 43 008 rows at 1024 x 1024.  Forward and input gradient are
``gemm3`` (fp32 operands split into bf16 pieces, six products, fp32-class error: tests/test_gemm3_gpu.py), the weight
gradient is its split-over-rows "NT" form whose staging loop also yields the bias gradient.  Shapes outside the kernels'
(in / out features not multiples of 32, fewer than 1 024 rows, non-fp32 or non-contiguous weight / bias, CPU) take ``F.linear``."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .gemm3 import gemm3, gemm3_nt, nt_reduce, split_weights_grouped


class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x2, weight, bias):
        pf, pb = split_weights_grouped([([weight], False), ([weight], True)])
        ctx.save_for_backward(x2, pb)
        ctx.has_bias = bias is not None
        return gemm3(x2, pf, bias)

    @staticmethod
    def backward(ctx, gy):
        from .encoder_fused import _balanced_rps
        x2, pb = ctx.saved_tensors
        g2 = gy if (gy.stride(1) == 1 and gy.stride(0) % 4 == 0) else gy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm3(g2, pb)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            rps = _balanced_rps(g2.shape[0], g2.shape[1], x2.shape[1], g2.device)
            c, ca, _ = gemm3_nt(g2, x2, rps, want_csum_a=True)
            dw, db = nt_reduce(c, ca)
            if not ctx.has_bias:
                db = None
        return dx, dw, db


def native_ok(x, weight, bias=None):
    """Both directions must fit the kernels: the forward contracts over in_features, the input gradient (dx = g @ W) over
    out_features — each a multiple of 32 (mpf_gemm3_tn's K step; ADVICE r5: out % 4 alone let a 72-wide Linear through the
    forward and fail in backward).  Dense fp32 weight, dense fp32 bias."""
    rows = x.numel() // max(x.shape[-1], 1)
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()
            and rows >= 1024 and x.shape[-1] % 32 == 0 and weight.shape[0] % 32 == 0
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.device == x.device))
            and not torch.is_autocast_enabled())


def linear_tall(x, weight, bias=None):
    """F.linear for inputs with many rows: the native fp32 GEMM when its shapes apply."""
    if not native_ok(x, weight, bias):
        return F.linear(x, weight, bias)
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1 or x2.stride(0) % 4 != 0:
        x2 = x2.contiguous()
    return _LinearFn.apply(x2, weight, bias).view(*x.shape[:-1], weight.shape[0])
