"""Channel-last GroupNorm kernels (csrc/groupnorm_cl.hip) against torch in fp64: plain, fused ReLU, fused FPN
top-down sum (norm(lateral) + bilinear 2x upsampling of the top map, msdeformattn.py:349), forward and gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _planes(N, C, H, W, dev, batch_pad=0, scale=2.0, shift=0.5):
    """[N, C, H, W] view over dense [H*W, C] planes; batch_pad > 0 leaves a gap between images (a level of the
    encoder memory looks like that)."""
    buf = torch.randn(N, H * W * C + batch_pad, device=dev) * scale + shift
    return buf[:, :H * W * C].view(N, H, W, C).permute(0, 3, 1, 2)


def _ref(x, w, b, groups, eps, relu, top):
    y = F.group_norm(x.double(), groups, w.double(), b.double(), eps)
    if relu:
        y = F.relu(y)
    if top is not None:
        y = y + F.interpolate(top.double(), size=y.shape[-2:], mode="bilinear", align_corners=False)
    return y


@pytest.mark.parametrize("shape,mode,pad", [
    ((2, 256, 64, 64), "plain", 0), ((2, 256, 64, 64), "relu", 0), ((2, 256, 64, 64), "top", 0),
    ((1, 256, 20, 28), "plain", 0), ((1, 256, 20, 28), "top", 512), ((3, 256, 9, 5), "relu", 256),
    ((2, 128, 34, 18), "top", 0), ((2, 64, 7, 3), "plain", 64), ((1, 256, 2, 2), "top", 0),
])
def test_group_norm_channel_last(shape, mode, pad):
    from mp_former_amd import _lib
    from mp_former_amd.groupnorm import GroupNorm, is_cl_plane
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(shape))
    N, C, H, W = shape
    G = 32 if C >= 128 else 16
    gn = GroupNorm(G, C).to(dev)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.normal_()
    x = _planes(N, C, H, W, dev, pad).requires_grad_(True)
    top = _planes(N, C, H // 2, W // 2, dev, pad).requires_grad_(True) if mode == "top" else None
    assert gn.cl_ok(x, top)
    y = gn.forward_cl(x, relu=mode == "relu", top=top)
    assert "gn_cl_apply" in _lib.last_kernel()
    assert is_cl_plane(y)
    xr = x.detach().clone().requires_grad_(True)
    tr = top.detach().clone().requires_grad_(True) if top is not None else None
    wr, br = gn.weight.detach().clone().requires_grad_(True), gn.bias.detach().clone().requires_grad_(True)
    yr = _ref(xr, wr, br, G, gn.eps, mode == "relu", tr)
    torch.testing.assert_close(y.double(), yr, rtol=2e-5, atol=2e-5)
    g = torch.randn(shape, device=dev)          # NCHW-contiguous gradient: the backward relayouts it itself
    if mode == "relu":
        # keep the comparison away from the kink: fp32 and fp64 may disagree on the sign of a ~0 output
        g = g * (yr.detach().abs() > 1e-4).float()
        g = torch.where((_ref(xr, wr, br, G, gn.eps, False, None).detach().abs() > 1e-4), g, torch.zeros_like(g))
    y.backward(g)
    yr.backward(g.double())
    scale = float(xr.grad.abs().max())
    torch.testing.assert_close(x.grad.double(), xr.grad.double(), rtol=1e-4, atol=2e-5 * max(scale, 1.0))
    torch.testing.assert_close(gn.weight.grad.double(), wr.grad.double(), rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))
    torch.testing.assert_close(gn.bias.grad.double(), br.grad.double(), rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))
    if top is not None:
        torch.testing.assert_close(top.grad.double(), tr.grad.double(), rtol=1e-5, atol=1e-5)


def test_group_norm_module_routes_channel_last_and_matches_nchw():
    """The module takes the channel-last kernels for conv-style outputs and the NCHW path for contiguous input; both
    agree with each other; a large-mean input exercises the Chan merge."""
    from mp_former_amd import _lib
    from mp_former_amd.groupnorm import GroupNorm
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    gn = GroupNorm(32, 256).to(dev)
    x = (torch.randn(2, 256, 96, 80, device=dev) * 0.3 + 40.0)
    xc = x.contiguous(memory_format=torch.channels_last)
    y_cl = gn(xc)
    assert "gn_cl_apply" in _lib.last_kernel()
    y_nchw = gn(x)
    ref = F.group_norm(x.double(), 32, gn.weight.double(), gn.bias.double(), gn.eps)
    torch.testing.assert_close(y_cl.double(), ref, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(y_nchw.double(), ref, rtol=1e-4, atol=1e-3)


def test_group_norm_flatten_equals_per_level_norm_and_cat():
    """group_norm_flatten: input_proj GroupNorm of three levels written straight into the encoder's [N, S, C] input
    (msdeformattn.py:319-322, :60-66) == per-level GroupNorm + flatten(2).transpose(1, 2) + cat, values and gradients."""
    from mp_former_amd.groupnorm import GroupNorm, group_norm_flatten
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    N, C = 2, 256
    sizes = [(8, 8), (16, 12), (32, 24)]
    norms, xs = [], []
    for i, (H, W) in enumerate(sizes):
        gn = GroupNorm(32, C).to(dev)
        with torch.no_grad():
            gn.weight.uniform_(0.5, 1.5); gn.bias.normal_()
        norms.append(gn)
        xs.append(_planes(N, C, H, W, dev, batch_pad=256 * i).detach().requires_grad_(True))
    y = group_norm_flatten(norms, xs)
    assert y is not None and y.shape == (N, sum(h * w for h, w in sizes), C) and y.is_contiguous()
    g = torch.randn_like(y)
    y.backward(g)
    got = [(x.grad.clone(), n.weight.grad.clone(), n.bias.grad.clone()) for x, n in zip(xs, norms)]
    for x, n in zip(xs, norms):
        x.grad = None; n.weight.grad = None; n.bias.grad = None
    y2 = torch.cat([n.forward_cl(x).flatten(2).transpose(1, 2) for n, x in zip(norms, xs)], 1)
    assert torch.equal(y, y2)
    y2.backward(g)
    for (dx, dw, db), x, n in zip(got, xs, norms):
        assert torch.equal(dx, x.grad) and torch.equal(dw, n.weight.grad) and torch.equal(db, n.bias.grad)
    ref = torch.cat([F.group_norm(x.detach().double(), 32, n.weight.double(), n.bias.double(), n.eps).flatten(2).transpose(1, 2)
                     for n, x in zip(norms, xs)], 1)
    assert (y.double() - ref).abs().max() < 1e-4
    # a level that does not qualify (NCHW-contiguous input) -> None: the caller keeps the per-level route
    assert group_norm_flatten(norms, [xs[0].detach().contiguous(), xs[1], xs[2]]) is None
