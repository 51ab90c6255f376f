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
    from mp_former_amd import _lib
    out = RefStyleFunction.apply(v, shapes.to(dev), lsi.to(dev), lo, a, 128)
    # a caller that never heard of attach_host_shapes lands on the PRODUCTION kernels
    assert "msda_fwd_block_kernel" in _lib.last_kernel(), _lib.last_kernel()
    out.backward(go.to(dev))
    assert "bin+tile" in _lib.last_kernel(), _lib.last_kernel()
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


def _problem(lv, N, Lq, seed, lsi=None, S=None, spread=1.2):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(lv, dtype=torch.long)
    run = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    lsi = run if lsi is None else torch.tensor(lsi, dtype=torch.long)
    S = int(shapes.prod(1).sum()) if S is None else S
    M, D, L, P = 8, 32, len(lv), 4
    Lq = S if Lq is None else Lq
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * spread - (spread - 1) / 2
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g)
    return value, shapes, lsi, loc, attn, go


def _smooth(loc, shapes):
    x = loc.double().numpy()[..., 0] * shapes.double().numpy()[None, None, None, :, None, 1] - 0.5
    y = loc.double().numpy()[..., 1] * shapes.double().numpy()[None, None, None, :, None, 0] - 0.5
    return ~((np.abs(x - np.round(x)) < 1e-4) | (np.abs(y - np.round(y)) < 1e-4))


@pytest.mark.parametrize("case", [
    dict(lv=[(8, 8), (16, 16), (32, 32)], N=2, Lq=None),                       # the pixel decoder's own shape class (queries = pixels)
    dict(lv=[(5, 7), (10, 14), (20, 28)], N=1, Lq=None),                       # sides that are multiples of neither 8 nor 4
    dict(lv=[(1, 50), (50, 1), (3, 3), (2, 9)], N=1, Lq=None),                 # thin maps: the worst case of the tile / block bounds
    dict(lv=[(16, 16)], N=2, Lq=100),                                          # one level, decoder-style queries (not the pixels)
    dict(lv=[(6, 4), (12, 9), (3, 2)], N=2, Lq=37),                            # the reference test's shape
    dict(lv=[(6, 4), (12, 9), (3, 2)], N=1, Lq=None, lsi=[5, 40, 150], S=170),  # levels NOT back to back: gaps before, between, after
    dict(lv=[(9, 9), (4, 4)], N=1, Lq=97, lsi=[16, 0], S=97),                  # levels stored in the other order
])
def test_device_geometry_route_matches_the_oracle(oracle_msda, case):
    """mpf_msda_forward_dev / mpf_msda_backward_dev — the blocked forward and the bin + tile backward on a geometry built ON THE
    DEVICE from spatial_shapes / level_start_index (ops/src/ms_deform_attn.h:25-66: the reference's op gets device tensors only) —
    against the C oracle (ops/functions/ms_deform_attn_func.py:52-72 restated) on shapes the launch-size estimates fit and on
    shapes they do not (thin maps, odd sides: the kernels stride), with level_start_index as the running sum and not."""
    from mp_former_amd import _lib, ms_deform_attn_backward, ms_deform_attn_forward
    dev = torch.device("cuda:0")
    value, shapes, lsi, loc, attn, go = _problem(case["lv"], case["N"], case["Lq"], 11, case.get("lsi"), case.get("S"))
    dv = [t.to(dev) for t in (value, shapes, lsi, loc, attn, go)]
    out = ms_deform_attn_forward(dv[0], dv[1], dv[2], dv[3], dv[4], 128)
    assert _lib.last_kernel() == "msda_fwd_block_kernel<dev>", _lib.last_kernel()
    gv, gl, ga = ms_deform_attn_backward(dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], 128)
    assert _lib.last_kernel() == "msda_bwd_block(bin+tile)<dev>", _lib.last_kernel()
    ref = oracle_msda.msda_forward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), attn.numpy())
    rgv, rgl, rga = oracle_msda.msda_backward(value.numpy(), shapes.numpy(), lsi.numpy(), loc.numpy(), attn.numpy(), go.numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gv.cpu().numpy(), rgv, rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(ga.cpu().numpy(), rga, rtol=1e-4, atol=1e-4)
    ok = _smooth(loc, shapes)
    np.testing.assert_allclose(gl.cpu().numpy()[ok], rgl[ok], rtol=1e-3, atol=2e-3)


def test_device_geometry_equals_the_host_geometry_route():
    """same kernels, same geometry: with level_start_index the running sum, the device-built record reproduces what the host builds
    from a copy of the shapes — the forward is BIT-identical, grad_loc / grad_attn too (one owner per sample), grad_value to
    summation order (the entries of a tile arrive in another order)."""
    from mp_former_amd import _lib, ms_deform_attn_backward, ms_deform_attn_forward
    from mp_former_amd.msda import attach_host_shapes
    dev = torch.device("cuda:0")
    lv = [(16, 24), (32, 48), (64, 96)]
    value, shapes, lsi, loc, attn, go = _problem(lv, 2, None, 5, spread=1.05)
    dv = [t.to(dev) for t in (value, shapes, lsi, loc, attn, go)]
    hs = attach_host_shapes(dv[1].clone(), lv, dv[2])
    o_h = ms_deform_attn_forward(dv[0], hs, dv[2], dv[3], dv[4], 128)
    assert _lib.last_kernel() == "msda_fwd_block_kernel", _lib.last_kernel()
    g_h = ms_deform_attn_backward(dv[0], hs, dv[2], dv[3], dv[4], dv[5], 128)
    assert _lib.last_kernel() == "msda_bwd_block(bin+tile)", _lib.last_kernel()
    o_d = ms_deform_attn_forward(dv[0], dv[1], dv[2], dv[3], dv[4], 128)
    g_d = ms_deform_attn_backward(dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], 128)
    assert _lib.last_kernel() == "msda_bwd_block(bin+tile)<dev>"
    assert torch.equal(o_h, o_d)
    assert torch.equal(g_h[1], g_d[1]) and torch.equal(g_h[2], g_d[2])
    torch.testing.assert_close(g_h[0], g_d[0], rtol=1e-5, atol=1e-5)


def test_device_geometry_route_never_synchronises():
    """forward + backward through the reference's call pattern with torch's synchronisation detector set to raise: no `.cpu()`,
    no `.item()`, no blocking copy anywhere on the route (the backward used to copy spatial_shapes to the host on every call)."""
    from mp_former_amd.msda import MSDeformAttnFunction
    dev = torch.device("cuda:0")
    value, shapes, lsi, loc, attn, go = _problem([(8, 8), (16, 16), (32, 32)], 2, None, 3)
    dv = [t.to(dev) for t in (value, shapes, lsi, loc, attn, go)]
    v, lo, a = (t.requires_grad_(True) for t in (dv[0], dv[3], dv[4]))
    MSDeformAttnFunction.apply(v, dv[1], dv[2], lo, a, 128).backward(dv[5])          # warm-up: workspace allocation
    v.grad = lo.grad = a.grad = None
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = MSDeformAttnFunction.apply(v, dv[1], dv[2], lo, a, 128)
        out.backward(dv[5])
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bool(torch.isfinite(v.grad).all()) and bool(torch.isfinite(lo.grad).all())


def test_device_geometry_rejects_overlapping_levels_with_nan_outputs():
    """shapes the prologue cannot serve — two levels claiming the same value rows — must not produce a plausible result: every output
    element is NaN and the geometry record says so (no host-visible error exists without a synchronisation)."""
    import ctypes
    from mp_former_amd import _lib, ms_deform_attn_backward, ms_deform_attn_forward
    from mp_former_amd import msda as msda_mod
    dev = torch.device("cuda:0")
    value, shapes, lsi, loc, attn, go = _problem([(6, 4), (12, 9)], 1, None, 9, lsi=[0, 10], S=132)
    dv = [t.to(dev) for t in (value, shapes, lsi, loc, attn, go)]
    out = ms_deform_attn_forward(dv[0], dv[1], dv[2], dv[3], dv[4], 128)
    gv, gl, ga = ms_deform_attn_backward(dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], 128)
    assert bool(torch.isnan(out).all()) and bool(torch.isnan(gv).all()) and bool(torch.isnan(gl).all()) and bool(torch.isnan(ga).all())
    ws = msda_mod._workspace(dev, 1024)
    rec = (ctypes.c_int * 10)()
    assert _lib.lib().mpf_msda_dev_geometry(ws.data_ptr(), rec, 10) == 0
    assert rec[0] == 0
    # ... and a valid call right after is served normally (nothing sticky)
    value, shapes, lsi, loc, attn, go = _problem([(6, 4), (12, 9)], 1, None, 9)
    dv = [t.to(dev) for t in (value, shapes, lsi, loc, attn, go)]
    out = ms_deform_attn_forward(dv[0], dv[1], dv[2], dv[3], dv[4], 128)
    assert bool(torch.isfinite(out).all())
    assert _lib.lib().mpf_msda_dev_geometry(msda_mod._workspace(dev, 1024).data_ptr(), rec, 10) == 0 and rec[0] == 1 and rec[1] == 1


def test_device_geometry_full_size_config_B_properties():
    """config B size (S = Lq = 21 504, N = 2): the device-geometry route against the host-geometry route of the same kernels —
    forward bit-identical, backward to summation order — and the launch estimate covers the real workgroup counts (no striding
    on the shapes the path was built for: record[2] / [3] <= the launched sizes)."""
    import ctypes
    from mp_former_amd import _lib, ms_deform_attn_backward, ms_deform_attn_forward
    from mp_former_amd import msda as msda_mod
    dev = torch.device("cuda:0")
    lv = [(128, 128), (64, 64), (32, 32)]
    g = torch.Generator().manual_seed(2)
    shapes = torch.tensor(lv, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, N, M, D, L, P = int(shapes.prod(1).sum()), 2, 8, 32, 3, 4
    value = torch.randn(N, S, M, D, generator=g).to(dev)
    loc = (torch.rand(N, S, M, L, P, 2, generator=g) * 1.1 - 0.05).to(dev)
    attn = torch.softmax(torch.randn(N, S, M, L * P, generator=g), -1).view(N, S, M, L, P).to(dev)
    go = torch.randn(N, S, M * D, generator=g).to(dev)
    sd, ld = shapes.to(dev), lsi.to(dev)
    hs = msda_mod.attach_host_shapes(sd.clone(), lv, ld)
    o_h = ms_deform_attn_forward(value, hs, ld, loc, attn, 128)
    g_h = ms_deform_attn_backward(value, hs, ld, loc, attn, go, 128)
    o_d = ms_deform_attn_forward(value, sd, ld, loc, attn, 128)
    g_d = ms_deform_attn_backward(value, sd, ld, loc, attn, go, 128)
    assert _lib.last_kernel() == "msda_bwd_block(bin+tile)<dev>"
    assert torch.equal(o_h, o_d) and torch.equal(g_h[1], g_d[1]) and torch.equal(g_h[2], g_d[2])
    torch.testing.assert_close(g_h[0], g_d[0], rtol=1e-4, atol=1e-4)
    rec = (ctypes.c_int * 10)()
    need = _lib.lib().mpf_msda_dev_workspace_bytes(N, S, M, L, S, P, 1)
    assert _lib.lib().mpf_msda_dev_geometry(msda_mod._workspace(dev, need).data_ptr(), rec, 10) == 0
    assert rec[0] == 1 and rec[1] == 1
    assert rec[2] == N * M * (256 + 64 + 16) and rec[3] == N * M * 448, list(rec)
