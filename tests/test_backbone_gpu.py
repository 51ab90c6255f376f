"""Bench backbone pieces: the fused shift/residual/ReLU epilogue (csrc/elementwise.hip) against the torch
ops it replaces (bit-exact: same rounding points of the activation), and a folded bottleneck against
conv -> FrozenBN -> ReLU composed from torch ops (fp32), forward and gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (False, False)])
def test_bias_act_matches_torch(dtype, with_res, relu):
    from mp_former_amd.backbone import bias_act
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(2, 64, 9, 7, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = torch.randn(2, 64, 9, 7, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    shift = torch.randn(64, device=dev)
    y = bias_act(x, shift, res, relu)
    # the shift stays fp32 inside the kernel (the torch composition would round it to bf16 first);
    # the activation is rounded where the unfused ops round it
    ref = (x.detach().float() + shift.view(1, -1, 1, 1)).to(dtype)
    if with_res:
        ref = (ref.float() + res.detach().float()).to(dtype)
    ref = F.relu(ref) if relu else ref
    assert torch.equal(y, ref)
    g = torch.randn_like(y)
    y.backward(g)
    gref = g * (ref > 0) if relu else g
    assert torch.equal(x.grad, gref)
    if with_res:
        assert torch.equal(res.grad, gref)


def test_folded_bottleneck_matches_unfolded_fp32():
    from mp_former_amd.backbone import Bottleneck
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    blk = Bottleneck(64, 32, 128, 2).to(dev)
    for m in blk.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 2.0); m.running_mean.normal_(); m.weight.uniform_(0.5, 1.5); m.bias.normal_()
    x = torch.randn(2, 64, 16, 16, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = blk(x)
    y.sum().backward()
    gx = x.grad.clone()
    gw = {n: p.grad.clone() for n, p in blk.named_parameters()}
    x.grad = None
    for p in blk.parameters():
        p.grad = None

    def bn(m, t):
        return F.batch_norm(t, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)

    o = F.relu(bn(blk.norm1, F.conv2d(x, blk.conv1.weight)))
    o = F.relu(bn(blk.norm2, F.conv2d(o, blk.conv2.weight, None, 2, 1)))
    o = bn(blk.norm3, F.conv2d(o, blk.conv3.weight))
    ref = F.relu(o + bn(blk.shortcut_norm, F.conv2d(x, blk.shortcut.weight, None, 2)))
    ref.sum().backward()
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gx, x.grad, rtol=1e-3, atol=1e-3)
    for n, p in blk.named_parameters():
        torch.testing.assert_close(gw[n], p.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_grouped_fold_cast_matches_foreach(channels_last, dtype):
    """_FoldCast (one native launch per direction) == w * scale[:, None, None, None] cast to the autocast dtype,
    and its backward == grad.float() * scale"""
    from mp_former_amd.backbone import _FoldCast
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(64, 3, 7, 7), (64, 64, 1, 1), (256, 64, 3, 3), (5, 7, 3, 3), (2048, 512, 1, 1)]
    ws = [torch.randn(s, device=dev, requires_grad=True) for s in shapes]
    if channels_last:
        ws = [w.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True) for w in ws]
    scales = [torch.rand(s[0], device=dev) + 0.5 for s in shapes]
    outs = _FoldCast.apply(dtype, scales, *ws)
    gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gs)
    for w, sc, o, g in zip(ws, scales, outs, gs):
        ref = w.detach() * sc.view(-1, 1, 1, 1)
        if dtype is not None:
            ref = ref.to(dtype)
        assert o.dtype == (dtype or torch.float32) and o.shape == w.shape and o.stride() == w.stride()
        assert torch.equal(o, ref)
        assert torch.equal(w.grad, g.float() * sc.view(-1, 1, 1, 1))


def test_forked_block_outputs_equal_the_unfused_graph_bitwise():
    """A stage of bottlenecks under bf16 autocast with the block outputs as two aliases (their gradients summed inside the
    ReLU-backward pass, csrc/elementwise.hip relu_bwd_add) against the same stage with one output per block (autograd's add
    kernel + threshold_backward): bit-identical outputs and input gradient, weight gradients to MIOpen's own run-to-run noise;
    plus the kernel alone."""
    import mp_former_amd.backbone as bb
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    stage = torch.nn.Sequential(bb.Bottleneck(64, 32, 128, 2), bb.Bottleneck(128, 32, 128, 1), bb.Bottleneck(128, 32, 128, 1)).to(dev)
    stage.to(memory_format=torch.channels_last)
    for m in stage.modules():
        if hasattr(m, "running_var"):
            m.running_var.uniform_(0.5, 2.0); m.running_mean.normal_(); m.weight.uniform_(0.5, 1.5); m.bias.normal_()
    x0 = torch.randn(2, 64, 24, 20, device=dev).contiguous(memory_format=torch.channels_last)
    extra = torch.randn(2, 128, 12, 10, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

    def run():
        x = x0.clone().requires_grad_(True)
        for p in stage.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = bb.run_stages(x.bfloat16(), [("s", stage)])["s"]
        # two consumers of the stage output, as in the model (next stage + pixel decoder)
        ((y.float() * y.float()).sum() + (y * extra).float().sum()).backward()
        return y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in stage.named_parameters()}

    fused = run()
    orig = bb.bias_act_fork
    bb.bias_act_fork = lambda x, shift, res: (lambda y: (y, y))(bb.bias_act(x, shift, res, True))
    try:
        plain = run()
    finally:
        bb.bias_act_fork = orig
    assert torch.equal(fused[0], plain[0]) and torch.equal(fused[1], plain[1])
    for n in fused[2]:      # (MIOpen's split-K weight-gradient kernels add with atomics: not bit-reproducible run to run)
        torch.testing.assert_close(fused[2][n], plain[2][n], rtol=2e-3, atol=2e-3 * float(plain[2][n].abs().max()))
    ga, gb = (torch.randn(2, 128, 12, 10, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) for _ in range(2))
    y = fused[0]
    assert torch.equal(bb._relu_bwd_add(ga, gb, y), torch.ops.aten.threshold_backward(ga + gb, y, 0))
    assert torch.equal(bb._relu_bwd_add(ga, None, y), torch.ops.aten.threshold_backward(ga, y, 0))


@pytest.mark.parametrize("shape", [(2, 64, 64, 48), (1, 8, 7, 9), (2, 16, 5, 5), (1, 64, 33, 32)])
def test_native_maxpool_equals_aten_bitwise(shape):
    """max_pool_3x3_s2 (csrc/elementwise.hip) == F.max_pool2d(x, 3, 2, 1) on channel-last bf16, forward and backward, including
    the tie rule: post-ReLU inputs (runs of equal zeros) and a NaN"""
    from mp_former_amd.backbone import max_pool_3x3_s2
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(shape))
    x = torch.relu(torch.randn(shape, device=dev)).bfloat16().contiguous(memory_format=torch.channels_last)
    x[0, 0, 1, 1] = float("nan")
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya, yb = max_pool_3x3_s2(a), F.max_pool2d(b, 3, 2, 1)
    assert ya.shape == yb.shape and ya.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(torch.nan_to_num(ya.float(), nan=-7.0), torch.nan_to_num(yb.float(), nan=-7.0))
    g = torch.randn_like(yb)
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(a.grad, b.grad)
