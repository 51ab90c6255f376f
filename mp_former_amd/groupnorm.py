"""nn.GroupNorm of the pixel decoder (msdeformattn.py:245-281) with chip-filling statistics.

torch computes the per-(image, group) moments with one workgroup per row; at batch 2 x 32 groups that is
64 workgroups on 256 CUs (216 us for the 134 MB maps at 256x256).  Here the statistics come from
``mpf_group_stats`` (rows cut into chunks, Chan merge), the affine apply is one element-wise pass, and the
backward is aten's ``native_group_norm_backward`` on the saved mean / rstd.  Same parameters and state-dict
keys as ``nn.GroupNorm``; other devices / dtypes / layouts take the stock implementation."""
import os

import torch
from torch import nn
from torch.autograd import Function

from . import _lib

_ws = {}


def _workspace(device, nbytes):
    w = _ws.get(device)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes) + 1024, dtype=torch.uint8, device=device)
        _ws[device] = w
    return w


def _transpose(src, B, R, C, in_bs=None):
    out = torch.empty(B * R * C, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        code = _lib.lib().mpf_transpose_f32(src.data_ptr(), R * C if in_bs is None else in_bs, out.data_ptr(), R * C, B, R, C,
                                            torch.cuda.current_stream(src.device).cuda_stream)
    _lib.check(code, "mpf_transpose_f32")
    return out


class _ToNCHW(Function):
    """channels_last [N, C, H, W] -> contiguous NCHW (and back for the gradient) with the LDS-tiled native
    transpose instead of aten's strided copy (150-180 us for a 134 MB map; ~50 us here)."""

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        ctx.dims = (N, C, H, W)
        return _transpose(x, N, H * W, C, x.stride(0)).view(N, C, H, W)

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.dims
        g = g.contiguous()
        return _transpose(g, N, C, H * W).view(N, H, W, C).permute(0, 3, 1, 2)


def to_nchw(x):
    # channel-last planes (unit channel stride, dense [H*W, C] per image; the batch stride is free: a level of the
    # encoder memory is such a view)
    if (x.dim() == 4 and x.dtype == torch.float32 and x.is_cuda and x.stride(1) == 1 and x.stride(3) == x.shape[1]
            and x.stride(2) == x.shape[3] * x.shape[1] and x.stride(0) % 4 == 0
            and not x.is_contiguous() and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.data_ptr() % 16 == 0
            and os.environ.get("MPF_FAST_TRANSPOSE", "1") == "1"):
        return _ToNCHW.apply(x)
    return x.contiguous()


class _GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        N, C = x.shape[:2]
        hw = x.numel() // (N * C)
        rows, row_len = N * groups, (C // groups) * hw
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        lib = _lib.lib()
        ws = _workspace(x.device, lib.mpf_group_stats_workspace_bytes(rows, row_len))
        with torch.cuda.device(x.device):
            code = lib.mpf_group_stats(x.data_ptr(), rows, row_len, float(eps), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(),
                                       ws.numel(), torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(code, "mpf_group_stats")
        # y = x * a + b with a[n, c] = rstd[n, g] * gamma[c], b[n, c] = beta[c] - mean[n, g] * a[n, c]
        a = (rstd.view(N, groups, 1) * weight.view(1, groups, C // groups)).view(N, C)
        b = (bias.view(1, groups, C // groups) - mean.view(N, groups, 1) * a.view(N, groups, C // groups)).view(N, C)
        shape = (N, C) + (1,) * (x.dim() - 2)
        y = torch.addcmul(b.view(shape), x, a.view(shape))
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.groups, ctx.hw = groups, hw
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd = ctx.saved_tensors
        N, C = x.shape[:2]
        gx, gw, gb = torch.ops.aten.native_group_norm_backward(
            gy.contiguous(), x, mean.view(N, ctx.groups), rstd.view(N, ctx.groups), weight, N, C, ctx.hw, ctx.groups,
            [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]])
        return gx, gw, gb, None, None


class GroupNorm(nn.GroupNorm):
    def forward(self, x):
        C = x.shape[1]
        if x.is_cuda and x.dtype == torch.float32 and not x.is_contiguous():
            x = to_nchw(x)          # channels_last conv outputs: aten's GroupNorm makes the same NCHW copy first
        if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and self.affine and x.dim() >= 3
                and ((C // self.num_groups) * (x.numel() // (x.shape[0] * C))) % 4 == 0):
            return _GroupNormFn.apply(x, self.weight, self.bias, self.num_groups, self.eps)
        return super().forward(x)
