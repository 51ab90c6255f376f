"""The 3x3 / stride 1 / padding 1 fp32 convolution of the FPN output stage (msdeformattn.py:272-281) on the split-bf16
GEMM: forward and input gradient are ONE GEMM each with K = 9 * Cin over channel-last planes (``mpf_gemm3_conv3x3``:
the A rows of a K step are read at the tap's (dy, dx) shift, taps off the image contribute zeros); the weight
gradient stays MIOpen's (``aten.convolution_backward``), which is already at the rate the NT kernel would reach.
Shapes outside the kernel (channel counts, alignment) keep the library convolution."""

import torch
from torch.autograd import Function

from . import _lib
from .gemm3 import amax, split_weights_grouped, split_weights_grouped_h2
from .groupnorm import is_cl_plane


def enabled():
    return True


def supported(x, weight):
    """fp32 channel-last planes that are dense across the batch, 3x3 kernel, channel counts the GEMM takes."""
    if not (enabled() and is_cl_plane(x) and weight.dtype == torch.float32 and weight.dim() == 4):
        return False
    N, C, H, W = x.shape
    Cout, Cin, kh, kw = weight.shape
    return (kh == 3 and kw == 3 and Cin == C and C % 32 == 0 and Cout % 32 == 0 and x.stride(0) == H * W * C
            and N * H * W * max(C, Cout) < 2 ** 31 and H >= 2 and W >= 2)


def _planes(N, C, H, W, device):
    return torch.empty((N, H, W, C), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


class _Conv3x3Fn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        # W2[co][(ky*3+kx)*Cin + ci] for the forward, W2t[ci][(ky*3+kx)*Cout + co] for the input gradient: one split launch
        w2 = weight.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
        w2t = weight.permute(1, 2, 3, 0).reshape(Cin, 9 * Cout)
        y = _planes(N, Cout, H, W, x.device)
        st = _lib.stream_ptr(x.device)
        h2 = Cout % 256 == 0 and Cin % 256 == 0          # the fp16 x 2 form (two-pass tiles both ways)
        if h2:
            (pf, pf_am), (pb, pb_am) = split_weights_grouped_h2([([w2.contiguous()], False), ([w2t.contiguous()], False)])
            x_am = amax(x.permute(0, 2, 3, 1))           # (the planes are dense in this order)
            with _lib.device_guard(x.device):
                code = _lib.lib().mpf_gemm3_conv3x3_h2(x.data_ptr(), x_am.data_ptr(), pf.data_ptr(), pf_am.data_ptr(),
                                                       bias.data_ptr() if bias is not None else None, y.data_ptr(), None,
                                                       N, H, W, Cin, Cout, 0, st)
            _lib.check(code, "mpf_gemm3_conv3x3_h2")
            ctx.save_for_backward(x, weight, pb, pb_am, x_am)
        else:
            pf, pb = split_weights_grouped([([w2.contiguous()], False), ([w2t.contiguous()], False)])
            with _lib.device_guard(x.device):
                code = _lib.lib().mpf_gemm3_conv3x3(x.data_ptr(), pf.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                    N, H, W, Cin, Cout, 0, st)
            _lib.check(code, "mpf_gemm3_conv3x3")
            ctx.save_for_backward(x, weight, pb)
        ctx.has_bias, ctx.h2 = bias is not None, h2
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.h2:
            x, weight, pb, pb_am, x_am = ctx.saved_tensors
        else:
            x, weight, pb = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        if not (is_cl_plane(gy) and gy.stride(0) == H * W * Cout):
            gy = gy.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        dx = dw = db = None
        st = _lib.stream_ptr(x.device)
        gy_am = amax(gy.permute(0, 2, 3, 1)) if ctx.h2 else None
        if ctx.needs_input_grad[0]:
            dx = _planes(N, Cin, H, W, x.device)
            with _lib.device_guard(x.device):
                if ctx.h2:
                    code = _lib.lib().mpf_gemm3_conv3x3_h2(gy.data_ptr(), gy_am.data_ptr(), pb.data_ptr(), pb_am.data_ptr(), None,
                                                           dx.data_ptr(), None, N, H, W, Cout, Cin, 1, st)
                else:
                    code = _lib.lib().mpf_gemm3_conv3x3(gy.data_ptr(), pb.data_ptr(), None, dx.data_ptr(), N, H, W, Cout, Cin, 1, st)
            _lib.check(code, "mpf_gemm3_conv3x3")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if W % 8 == 0 and Cin % 128 == 0:
                dw, db = _wgrad_native(gy, x, N, H, W, Cin, Cout, ctx.has_bias, (gy_am, x_am) if ctx.h2 else None)
            else:
                _, dw, db = torch.ops.aten.convolution_backward(
                    gy, x, weight, [Cout] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                    [False, bool(ctx.needs_input_grad[1]), bool(ctx.has_bias and ctx.needs_input_grad[2])])
        return dx, dw, db


def _wgrad_native(gy, x, N, H, W, Cin, Cout, has_bias, amax_ab=None):
    """dW [Cout, Cin, 3, 3] (as a channels_last-strided view of [Cout, 3, 3, Cin]) and the bias gradient on the split-bf16 NT
    kernel's convolution mode: one launch over (output tile, row split) + the fixed-order sum of the splits."""
    from .gemm3 import nt_reduce
    R = N * H * W
    tiles = ((Cout + 127) // 128) * (9 * Cin // 128)
    slots = 2 * torch.cuda.get_device_properties(x.device).multi_processor_count
    ns = max(1, (3 * slots) // tiles)                       # ~3 rounds of workgroups: measured best at 256 -> 256 channels
    if amax_ab is not None and Cout % 256 == 0 and Cin % 256 == 0:
        # 256 x 256 tiles (csrc/gemm3_nt2.h): one 8-wave workgroup per CU, one round
        ns = max(1, (slots // 2) // ((Cout // 256) * (9 * Cin // 256)))
    rps = max(128, ((-(-R // ns)) + 31) // 32 * 32)
    ns = -(-R // rps)
    c = torch.empty((ns, Cout, 9 * Cin), dtype=torch.float32, device=x.device)
    ca = torch.empty((ns, Cout), dtype=torch.float32, device=x.device) if has_bias else None
    with _lib.device_guard(x.device):
        if amax_ab is not None:
            code = _lib.lib().mpf_gemm3_conv3x3_wgrad_h2(gy.data_ptr(), amax_ab[0].data_ptr(), x.data_ptr(), amax_ab[1].data_ptr(), c.data_ptr(),
                                                         ca.data_ptr() if has_bias else None, N, H, W, Cin, Cout, rps,
                                                         _lib.stream_ptr(x.device))
        else:
            code = _lib.lib().mpf_gemm3_conv3x3_wgrad(gy.data_ptr(), x.data_ptr(), c.data_ptr(), ca.data_ptr() if has_bias else None, N, H, W,
                                                      Cin, Cout, rps, _lib.stream_ptr(x.device))
    _lib.check(code, "mpf_gemm3_conv3x3_wgrad")
    dw2, db = nt_reduce(c, ca)
    return dw2.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), db


def conv3x3(x, weight, bias=None):
    return _Conv3x3Fn.apply(x, weight, bias)


# ---------------------------------------------------------------------------------------------------------------------
# 1x1 convolutions of the pixel decoder (input projections, FPN lateral, mask_features: msdeformattn.py:245-262, :266-271,
# :284-291) on channel-last planes ARE matrix products [N*H*W, Cin] x [Cin, Cout]: forward and input gradient on
# mpf_gemm3_tn, weight + bias gradient on the split-over-rows mpf_gemm3_nt (bias gradient = its column sums).
# ---------------------------------------------------------------------------------------------------------------------
def _planes_any(x):
    """[N, C, H, W] fp32 / bf16 whose images are dense [H*W, C] planes, dense across the batch, 16-byte aligned rows"""
    if not (x.dim() == 4 and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    N, C, H, W = x.shape
    return (x.stride(1) == 1 and x.stride(3) == C and x.stride(2) == W * C and x.stride(0) == H * W * C
            and x.data_ptr() % 16 == 0 and (C * x.element_size()) % 16 == 0)


def supported_1x1(x, weight):
    if not (_planes_any(x) and weight.dtype == torch.float32 and weight.dim() == 4):
        return False
    N, C, H, W = x.shape
    Cout, Cin, kh, kw = weight.shape
    return kh == 1 and kw == 1 and Cin == C and C % 32 == 0 and Cout % 32 == 0


class _Conv1x1Fn(Function):
    """x may be bf16 (a backbone feature map under autocast: it enters the fp32 GEMM as its own first plane — no cast pass,
    three products instead of six — and receives a bf16 gradient straight from the GEMM epilogue); out_dtype bf16 makes the
    result bf16 (rounded once: what ``.to(bfloat16)`` of the fp32 result would give) and the incoming gradient is then
    consumed in bf16 as well."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype):
        from .gemm3 import gemm3, gemm3_ex
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        w2 = weight.view(Cout, Cin)
        pf, pb = split_weights_grouped([([w2], False), ([w2], True)])
        x2 = x.permute(0, 2, 3, 1).reshape(N * H * W, Cin)                      # view of the planes
        if x.dtype == torch.float32 and out_dtype == torch.float32:
            y2 = gemm3(x2, pf, bias)
        else:
            y2 = gemm3_ex(x2, pf, bias, out_dtype=out_dtype)
        ctx.save_for_backward(x2, pb)
        ctx.dims, ctx.has_bias = (N, Cin, Cout, H, W), bias is not None
        return y2.view(N, H, W, Cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from .encoder_fused import _balanced_rps
        from .gemm3 import gemm3, gemm3_ex, gemm3_nt, gemm3_nt_ex, nt_reduce
        x2, pb = ctx.saved_tensors
        N, Cin, Cout, H, W = ctx.dims
        g2 = gy.permute(0, 2, 3, 1).reshape(N * H * W, Cout)
        if g2.stride(1) != 1 or g2.stride(0) != Cout:
            g2 = g2.contiguous()
        if g2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16:
            g2 = g2.float()                                                     # (not a case of the pixel decoder)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if g2.dtype == torch.float32 and x2.dtype == torch.float32:
                dx2 = gemm3(g2, pb)
            else:
                dx2 = gemm3_ex(g2, pb, out_dtype=x2.dtype)                       # bf16 input -> bf16 gradient, no cast pass
            dx = dx2.view(N, H, W, Cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            rps = _balanced_rps(g2.shape[0], Cout, Cin, g2.device)
            mixed = g2.dtype != x2.dtype
            if mixed and Cin % 128 == 0:
                c, ca = gemm3_nt_ex(g2, x2, rps, want_csum_a=True)
            else:
                c, ca, _ = gemm3_nt(g2.float() if mixed else g2, x2.float() if mixed else x2, rps, want_csum_a=True)
            dw, db = nt_reduce(c, ca)
            dw = dw.view(Cout, Cin, 1, 1)
            if not ctx.has_bias:
                db = None
        return dx, dw, db, None


def conv1x1(x, weight, bias=None, out_dtype=torch.float32):
    return _Conv1x1Fn.apply(x, weight, bias, out_dtype)
