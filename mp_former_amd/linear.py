"""nn.Linear forward/backward with a split-K weight gradient for tall inputs.

The encoder's Linear layers see [N*S, C] activations with N*S = 43 008 rows at 1024x1024: the
weight-gradient GEMM  dW[out,in] = dY^T[out,rows] . X[rows,in]  has a tiny output (256x1024) and a
huge reduction dimension, which the library runs as a handful of workgroups (measured 43 TFLOP/s
fp32).  Splitting the rows into chunks turns it into a batched GEMM that fills the 256 CUs, followed
by a small sum.  Same math (fp32 accumulation, summation order differs)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function


def _pick_chunks(rows, limit=32):
    for c in (32, 24, 16, 12, 8, 6, 4, 3, 2):
        if c <= limit and rows % c == 0 and rows // c >= 1024:
            return c
    return 1


class _LinearSplitK(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        g2 = gy.reshape(-1, gy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        if ctx.needs_input_grad[0]:
            gx = (g2 @ weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            rows = g2.shape[0]
            c = _pick_chunks(rows)
            if c > 1:
                gw = torch.bmm(g2.view(c, rows // c, -1).transpose(1, 2), x2.view(c, rows // c, -1)).sum(0)
            else:
                gw = g2.t() @ x2
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


def linear_tall(x, weight, bias=None):
    """F.linear for inputs with many rows; identical forward, split-K weight gradient."""
    if x.numel() // x.shape[-1] < 8192 or not x.is_cuda:
        return F.linear(x, weight, bias)
    return _LinearSplitK.apply(x, weight, bias)
