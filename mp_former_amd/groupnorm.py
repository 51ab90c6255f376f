"""nn.GroupNorm of the pixel decoder (msdeformattn.py:245-281).

Channel-last planes (what MIOpen's fp32 convolutions return) take csrc/groupnorm_cl.hip: chunked statistics
with a Chan merge, ONE apply pass (optionally fused with the following ReLU or with the FPN top-down sum
``+ upsample2x(top)``) and a native backward, all without leaving the layout (shapes outside the kernels keep the library GroupNorm).
Contiguous NCHW input takes round 1's path: ``mpf_group_stats`` (rows cut into chunks; torch's
one-workgroup-per-row moments fill 64 of 256 CUs at batch 2), an element-wise apply, and aten's
``native_group_norm_backward``.  Same parameters and state-dict keys as ``nn.GroupNorm``; other devices /
dtypes take the stock implementation."""

import torch
from torch import nn
from torch.autograd import Function

from . import _lib

_ws = {}


def _workspace(device, nbytes):
    key = (device, _lib.ws_scope())          # (a graph capture has its own buffers: _lib.workspace_scope)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes) + 1024, dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


def _transpose(src, B, R, C, in_bs=None):
    out = torch.empty(B * R * C, dtype=torch.float32, device=src.device)
    with _lib.device_guard(src.device):
        code = _lib.lib().mpf_transpose_f32(src.data_ptr(), R * C if in_bs is None else in_bs, out.data_ptr(), R * C, B, R, C,
                                            _lib.stream_ptr(src.device))
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
            and not x.is_contiguous() and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.data_ptr() % 16 == 0):
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
        with _lib.device_guard(x.device):
            code = lib.mpf_group_stats(x.data_ptr(), rows, row_len, float(eps), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(),
                                       ws.numel(), _lib.stream_ptr(x.device))
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


def is_cl_plane(x):
    """[N, C, H, W] fp32 whose images are dense [H*W, C] planes (channels_last; the batch stride is free: a level of
    the encoder memory is such a view)."""
    return (x.dim() == 4 and x.dtype == torch.float32 and x.is_cuda and x.stride(1) == 1 and x.stride(3) == x.shape[1]
            and x.stride(2) == x.shape[3] * x.shape[1] and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def _cl_empty(N, C, H, W, device):
    return torch.empty((N, H, W, C), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


class _GroupNormCLFn(Function):
    """GroupNorm on channel-last planes (csrc/groupnorm_cl.hip): statistics + ONE apply pass, optionally fused with
    the ReLU of detectron2's Conv2d wrapper or with the FPN top-down sum ``+ upsample2x(top)``; the backward is
    two passes over (gy, x) plus a gather for the gradient of ``top``."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, top):
        N, C, H, W = x.shape
        lib = _lib.lib()
        mean = torch.empty(N * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N * groups, dtype=torch.float32, device=x.device)
        y = _cl_empty(N, C, H, W, x.device)
        ws = _workspace(x.device, lib.mpf_gn_cl_workspace_bytes(N, H * W, C, groups))
        with _lib.device_guard(x.device):
            code = lib.mpf_gn_cl_forward(x.data_ptr(), x.stride(0), weight.data_ptr(), bias.data_ptr(), N, H * W, C, groups, float(eps),
                                         1 if relu else 0, top.data_ptr() if top is not None else None,
                                         top.stride(0) if top is not None else 0, W, y.data_ptr(), y.stride(0), mean.data_ptr(),
                                         rstd.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device))
        _lib.check(code, "mpf_gn_cl_forward")
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.groups, ctx.relu, ctx.has_top = groups, bool(relu), top is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        if not is_cl_plane(gy):
            gy = gy.contiguous(memory_format=torch.channels_last)
            if not is_cl_plane(gy):          # N == 1 or C == 1: the memory-format query is ambiguous; force the planes
                gy = gy.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        lib = _lib.lib()
        dx = _cl_empty(N, C, H, W, x.device)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _workspace(x.device, lib.mpf_gn_cl_workspace_bytes(N, H * W, C, ctx.groups))
        stream = _lib.stream_ptr(x.device)
        dtop = None
        with _lib.device_guard(x.device):
            code = lib.mpf_gn_cl_backward(gy.data_ptr(), gy.stride(0), x.data_ptr(), x.stride(0), weight.data_ptr(), bias.data_ptr(),
                                          mean.data_ptr(), rstd.data_ptr(), N, H * W, C, ctx.groups, 1 if ctx.relu else 0,
                                          dx.data_ptr(), dx.stride(0), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), stream)
            _lib.check(code, "mpf_gn_cl_backward")
            if ctx.has_top and ctx.needs_input_grad[6]:
                dtop = _cl_empty(N, C, H // 2, W // 2, x.device)
                code = lib.mpf_upsample2x_cl_backward(gy.data_ptr(), gy.stride(0), N, H // 2, W // 2, C, dtop.data_ptr(),
                                                      dtop.stride(0), stream)
                _lib.check(code, "mpf_upsample2x_cl_backward")
        return dx, dg, db, None, None, None, dtop


class _GroupNormFlattenFn(Function):
    """GroupNorm of several channel-last maps written straight into ONE [N, sum(H_l W_l), C] buffer — the encoder's
    ``src_flatten`` (msdeformattn.py:319-322 + :60-66: input_proj GroupNorm, flatten(2).transpose(1, 2), cat over levels)
    without the per-level outputs and the concatenation pass.  args = (x_0, weight_0, bias_0, x_1, ...)."""

    @staticmethod
    def forward(ctx, groups, eps, *args):
        xs, ws, bs = args[0::3], args[1::3], args[2::3]
        N, C = xs[0].shape[:2]
        sizes = [int(x.shape[2]) * int(x.shape[3]) for x in xs]
        S = sum(sizes)
        dev = xs[0].device
        lib = _lib.lib()
        out = torch.empty((N, S, C), dtype=torch.float32, device=dev)
        stats = []
        stream = _lib.stream_ptr(dev)
        off = 0
        with _lib.device_guard(dev):
            for x, w, b, hw in zip(xs, ws, bs, sizes):
                mean = torch.empty(N * groups, dtype=torch.float32, device=dev)
                rstd = torch.empty(N * groups, dtype=torch.float32, device=dev)
                wsb = _workspace(dev, lib.mpf_gn_cl_workspace_bytes(N, hw, C, groups))
                code = lib.mpf_gn_cl_forward(x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(), N, hw, C, groups, float(eps), 0,
                                             None, 0, int(x.shape[3]), out.data_ptr() + off * C * 4, S * C, mean.data_ptr(),
                                             rstd.data_ptr(), wsb.data_ptr(), wsb.numel(), stream)
                _lib.check(code, "mpf_gn_cl_forward")
                stats += [mean, rstd]
                off += hw
        ctx.save_for_backward(*xs, *ws, *bs, *stats)
        ctx.groups, ctx.sizes, ctx.nl = groups, sizes, len(xs)
        return out

    @staticmethod
    def backward(ctx, g):
        nl = ctx.nl
        sv = ctx.saved_tensors
        xs, ws, bs, stats = sv[:nl], sv[nl:2 * nl], sv[2 * nl:3 * nl], sv[3 * nl:]
        g = g.contiguous()
        N, S, C = g.shape
        dev = g.device
        lib = _lib.lib()
        stream = _lib.stream_ptr(dev)
        grads = []
        off = 0
        with _lib.device_guard(dev):
            for l, (x, w, b, hw) in enumerate(zip(xs, ws, bs, ctx.sizes)):
                H, W = int(x.shape[2]), int(x.shape[3])
                dx = _cl_empty(N, C, H, W, dev)
                dg = torch.empty(C, dtype=torch.float32, device=dev)
                db = torch.empty(C, dtype=torch.float32, device=dev)
                wsb = _workspace(dev, lib.mpf_gn_cl_workspace_bytes(N, hw, C, ctx.groups))
                code = lib.mpf_gn_cl_backward(g.data_ptr() + off * C * 4, S * C, x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(),
                                              stats[2 * l].data_ptr(), stats[2 * l + 1].data_ptr(), N, hw, C, ctx.groups, 0,
                                              dx.data_ptr(), dx.stride(0), dg.data_ptr(), db.data_ptr(), wsb.data_ptr(), wsb.numel(),
                                              stream)
                _lib.check(code, "mpf_gn_cl_backward")
                grads += [dx, dg, db]
                off += hw
        return (None, None) + tuple(grads)


def group_norm_flatten(norms, xs):
    """[GroupNorm_l(x_l)] flattened and concatenated over levels -> [N, sum(HW_l), C] in one buffer; None if a level does
    not qualify for the channel-last kernels (the caller then takes the per-level route)."""
    g0 = norms[0]
    for n, x in zip(norms, xs):
        if not (isinstance(n, GroupNorm) and n.cl_ok(x) and n.num_groups == g0.num_groups and n.eps == g0.eps
                and x.shape[:2] == xs[0].shape[:2]):
            return None
    args = []
    for n, x in zip(norms, xs):
        args += [x, n.weight, n.bias]
    return _GroupNormFlattenFn.apply(g0.num_groups, g0.eps, *args)


def cl_enabled():
    return True


class GroupNorm(nn.GroupNorm):
    def cl_ok(self, x, top=None):
        """The channel-last kernels apply: fp32 planes, affine, supported (C, G); ``top`` (FPN top-down map) exactly
        half the size."""
        if not (cl_enabled() and self.affine and is_cl_plane(x) and self.weight.dtype == torch.float32):
            return False
        N, C, H, W = x.shape
        if not _lib.lib().mpf_gn_cl_supported(H * W, C, self.num_groups):
            return False
        if top is not None:
            return (is_cl_plane(top) and top.shape[0] == N and top.shape[1] == C and top.shape[2] * 2 == H
                    and top.shape[3] * 2 == W)
        return True

    def forward_cl(self, x, relu=False, top=None):
        return _GroupNormCLFn.apply(x, self.weight, self.bias, self.num_groups, self.eps, relu, top)

    def forward(self, x):
        C = x.shape[1]
        if self.cl_ok(x):
            return self.forward_cl(x)
        if x.is_cuda and x.dtype == torch.float32 and not x.is_contiguous():
            x = to_nchw(x)          # channels_last conv outputs: aten's GroupNorm makes the same NCHW copy first
        if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and self.affine and x.dim() >= 3
                and ((C // self.num_groups) * (x.numel() // (x.shape[0] * C))) % 4 == 0):
            return _GroupNormFn.apply(x, self.weight, self.bias, self.num_groups, self.eps)
        return super().forward(x)
