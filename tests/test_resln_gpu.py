"""Fused residual + LayerNorm (csrc/elementwise.hip mpf_res_ln256_*) against add + F.layer_norm (+ cast):
forward and all gradients, fp32 branch and bf16 branch (the AMP decoder), and the no-branch form
(decoder_norm)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", [3, 240, 2051])
@pytest.mark.parametrize("tdtype", [None, torch.float32, torch.bfloat16])
def test_res_ln_matches_torch(rows, tdtype):
    from mp_former_amd.resln import res_ln
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    norm = torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5); norm.bias.normal_()
    x = torch.randn(rows, 1, 256, device=dev, requires_grad=True)
    t = None if tdtype is None else torch.randn(rows, 1, 256, device=dev).to(tdtype).requires_grad_(True)
    amp = tdtype == torch.bfloat16
    y32, y16 = res_ln(norm, x, t, want32=True, want16=amp)
    xr = x.detach().clone().requires_grad_(True)
    tr = None if t is None else t.detach().clone().requires_grad_(True)
    s = xr if tr is None else xr + tr.float()
    r32 = F.layer_norm(s, (256,), norm.weight, norm.bias, norm.eps)
    torch.testing.assert_close(y32, r32, rtol=1e-5, atol=1e-5)
    g32 = torch.randn_like(y32)
    loss, rloss = (y32 * g32).sum(), (r32 * g32).sum()
    if amp:
        assert torch.equal(y16, y32.to(torch.bfloat16))
        g16 = torch.randn_like(y32).to(torch.bfloat16)
        loss = loss + (y16.float() * g16.float()).sum()
        rloss = rloss + (r32.to(torch.bfloat16).float() * g16.float()).sum()
    gw = torch.autograd.grad(loss, [x] + ([t] if t is not None else []) + [norm.weight, norm.bias])
    rw = torch.autograd.grad(rloss, [xr] + ([tr] if tr is not None else []) + [norm.weight, norm.bias])
    for a, b in zip(gw, rw):
        tol = 2e-2 if a.dtype == torch.bfloat16 else 2e-4
        torch.testing.assert_close(a.float(), b.float(), rtol=tol, atol=tol * (1 + float(b.float().abs().max())))


@pytest.mark.parametrize("shape", [(2, 256, 64, 64), (1, 256, 20, 28), (3, 64, 9, 4)])
def test_group_norm_matches_torch(shape):
    """chunked GroupNorm statistics (mpf_group_stats) + one-pass apply against nn.GroupNorm, forward and gradients."""
    from mp_former_amd.groupnorm import GroupNorm
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    gn = GroupNorm(32, shape[1]).to(dev)
    ref = torch.nn.GroupNorm(32, shape[1]).to(dev)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.normal_()
        ref.weight.copy_(gn.weight); ref.bias.copy_(gn.bias)
    x = (torch.randn(shape, device=dev) * 2 + 0.5).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    g = torch.randn(shape, device=dev)
    y, yr = gn(x), ref(xr)
    torch.testing.assert_close(y, yr, rtol=2e-5, atol=2e-5)
    y.backward(g); yr.backward(g)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gn.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(gn.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("shape", [(2, 256, 16, 24), (1, 8, 2, 2), (2, 256, 64, 64), (3, 36, 10, 6)])
def test_to_nchw_matches_contiguous(shape):
    from mp_former_amd.groupnorm import to_nchw
    torch.manual_seed(0)
    x = torch.randn(shape, device="cuda:0").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = to_nchw(x)
    assert y.is_contiguous() and torch.equal(y, x.detach().contiguous())
    g = torch.randn(shape, device="cuda:0")
    y.backward(g)
    assert torch.equal(x.grad, g)


def test_ln256_amax_slots_bound_forward_exact_backward():
    """The amax slots the fp16 x 2 GEMMs read (include/mpformer_hip.h): the forward writes an upper BOUND of |y| and |y + padd| from
    (gamma, beta, slot of padd) — never below the true maximum, exactly 16 max|gamma| + max|beta| (+ max|padd|) — the backward the
    exact max |ds|; results unchanged by the slot arguments."""
    from mp_former_amd.gemm3 import amax, amax_slots, amax_value
    from mp_former_amd.resln import ln256_backward, ln256_forward
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    rows = 5000
    x = torch.randn(rows, 256, device=dev) * 3 + 1
    g, b = torch.randn(256, device=dev), torch.randn(256, device=dev)
    padd = torch.randn(100, 256, device=dev) * 2
    sl = amax_slots(3, dev)
    pa = amax(padd)
    y0, m0, r0, yp0 = ln256_forward(x, g, b, 1e-5, padd=padd)
    y, m, r, yp = ln256_forward(x, g, b, 1e-5, padd=padd, y_bound=sl[0], padd_amax=pa, yplus_bound=sl[1])
    assert torch.equal(y, y0) and torch.equal(yp, yp0) and torch.equal(m, m0) and torch.equal(r, r0)
    bound = 16.0 * float(g.abs().max()) + float(b.abs().max())
    assert abs(float(amax_value(sl[0])) - bound) <= 1e-5 * bound and float(amax_value(sl[0])) >= float(y.abs().max())
    bp = bound + float(padd.abs().max())
    assert abs(float(amax_value(sl[1])) - bp) <= 1e-5 * bp and float(amax_value(sl[1])) >= float(yp.abs().max())
    gy = torch.randn(rows, 256, device=dev) * 1e-3
    ds0, dg0, db0 = ln256_backward(x, m, r, g, gy)
    ds, dg, db = ln256_backward(x, m, r, g, gy, ds_amax=sl[2])
    assert torch.equal(ds, ds0) and torch.equal(dg, dg0) and torch.equal(db, db0)
    assert float(amax_value(sl[2])) == float(ds.abs().max())


def test_ln_grad_group_equals_single_backwards():
    """LnGradGroup (several LayerNorm backwards, ONE parameter-gradient reduce launch: the encoder's twelve) == ln256_backward per
    LayerNorm, bit for bit (same per-workgroup partials, same fixed-order sums), with and without the second gradient operand and
    the amax slot."""
    from mp_former_amd.gemm3 import amax_slots, amax_value
    from mp_former_amd.resln import LnGradGroup, ln256_backward, ln256_forward
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    rows = 4300
    grp = LnGradGroup(3, rows, dev)
    want, got = [], []
    sl = amax_slots(3, dev)
    for z in range(3):
        x = torch.randn(rows, 256, device=dev) * (z + 1)
        g, b = torch.randn(256, device=dev), torch.randn(256, device=dev)
        _, m, r, _ = ln256_forward(x, g, b, 1e-5)
        gy = torch.randn(rows, 256, device=dev) * 1e-2
        gp = torch.randn(rows, 256, device=dev) * 1e-2 if z == 1 else None
        want.append(ln256_backward(x, m, r, g, gy, gp))
        got.append(grp.backward(x, m, r, g, gy, gp, ds_amax=sl[z] if z != 2 else None))
    dgb = grp.finish()
    for z in range(3):
        assert torch.equal(got[z], want[z][0])
        assert torch.equal(dgb[z, 0], want[z][1]) and torch.equal(dgb[z, 1], want[z][2])
    assert float(amax_value(sl[0])) == float(got[0].abs().max()) and float(amax_value(sl[1])) == float(got[1].abs().max())
