"""The drop-in import name.  The reference's ops/functions/ms_deform_attn_func.py:21-49 does

    import MultiScaleDeformableAttention as MSDA
    ...
    output = MSDA.ms_deform_attn_forward(value, shapes, level_start, loc, attn, im2col_step)
    grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(value, shapes, level_start, loc, attn, grad_output, im2col_step)

This test restates exactly that consumer (an autograd Function written against the module NAME and its two positional
signatures) after `mp_former_amd.dropin.install()`, and checks it against the C oracle."""
import importlib
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reference_import_pattern_binds_to_the_hip_kernels(oracle_msda):
    import mp_former_amd.dropin as dropin
    sys.modules.pop("MultiScaleDeformableAttention", None)
    dropin.install()
    MSDA = importlib.import_module("MultiScaleDeformableAttention")       # the reference's import (func.py:22)
    assert hasattr(MSDA, "ms_deform_attn_forward") and hasattr(MSDA, "ms_deform_attn_backward")

    class RefStyleFunction(torch.autograd.Function):                       # restatement of func.py:32-49's call pattern
        @staticmethod
        def forward(ctx, value, shapes, lsi, loc, attn, im2col_step):
            ctx.im2col_step = im2col_step
            out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, ctx.im2col_step)
            ctx.save_for_backward(value, shapes, lsi, loc, attn)
            return out

        @staticmethod
        def backward(ctx, grad_output):
            value, shapes, lsi, loc, attn = ctx.saved_tensors
            gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output.contiguous(), ctx.im2col_step)
            return gv, None, None, gl, ga, None

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    lv = [(6, 4), (12, 9), (3, 2)]
    shapes = torch.tensor(lv, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, N, M, D, L, P, Lq = int(shapes.prod(1).sum()), 2, 8, 32, 3, 4, 37
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g)
    v, lo, a = (t.to(dev).requires_grad_(True) for t in (value, loc, attn))
    out = RefStyleFunction.apply(v, shapes.to(dev), lsi.to(dev), lo, a, 128)
    out.backward(go.to(dev))
    ref = oracle_msda.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    rgv, rgl, rga = oracle_msda.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), attn.numpy(), go.numpy())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v.grad.cpu().numpy(), rgv, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(a.grad.cpu().numpy(), rga, rtol=1e-4, atol=1e-4)
    x = loc.double().numpy()[..., 0] * shapes.double().numpy()[None, None, None, :, None, 1] - 0.5
    y = loc.double().numpy()[..., 1] * shapes.double().numpy()[None, None, None, :, None, 0] - 0.5
    ok = ~((np.abs(x - np.round(x)) < 1e-4) | (np.abs(y - np.round(y)) < 1e-4))
    np.testing.assert_allclose(lo.grad.cpu().numpy()[ok], rgl[ok], rtol=1e-3, atol=2e-3)
    # CPU tensors raise like the reference's op (ms_deform_attn.h:43)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 128)
