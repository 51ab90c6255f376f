"""The natively sequenced decoder layer (csrc/decoder_layer.hip) against the op-by-op path: same kernels,
same rounding points — losses and gradients must agree to bf16 accumulation-order noise."""
import os

import pytest
import torch

from conftest import fifo_to_tags, load_head_fixture
from test_head_gpu import _build

pytestmark = pytest.mark.gpu


def _run(h, feats, targets, replay, cfg, fused, pin=None):
    """One AMP step on replayed draws.  ``pin``: a list shared by the runs that are compared with each other — the first run
    solves the assignment on the reference's route (cost matrices to the host, SciPy) and records it, the later ones get it
    back from the matcher.  The toy fixtures have near-duplicate queries whose costs tie to within bf16's rounding, and the
    two routes under comparison round differently: an assignment that flips between them is an O(0.1) change of that output's
    losses which says nothing about the kernels compared (the device solver is pinned to SciPy by tests/test_lsa_gpu.py)."""
    from mp_former_amd import _rng
    os.environ["MPF_FUSED_DECODER"] = "1" if fused else "0"
    h.zero_grad(set_to_none=True)
    tags = fifo_to_tags(replay, cfg, True)
    matcher = h.criterion.matcher
    solve = matcher.match_many
    if pin is not None:
        os.environ["MPF_DEVICE_LSA"] = "0"
        if pin:
            matcher.match_many = lambda *a, **k: pin[0]
            tags = {t: d for t, d in tags.items() if not t.startswith("match")}     # (the matcher's own draws are not consumed)
        else:
            matcher.match_many = lambda *a, **k: pin.append(solve(*a, **k)) or pin[-1]
    _rng.install_replay(tags)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            losses, _ = h(feats, targets)
            total = sum(losses.values())
        total.backward()
    finally:
        _rng.install_replay(None)
        os.environ.pop("MPF_FUSED_DECODER", None)
        if pin is not None:
            matcher.match_many = solve
            os.environ.pop("MPF_DEVICE_LSA", None)
    grads = {n: p.grad.detach().clone() for n, p in h.named_parameters() if p.grad is not None}
    return {k: float(v) for k, v in losses.items()}, grads


def test_fused_decoder_layer_matches_op_by_op_path():
    from mp_former_amd import _lib
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_small")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    pin = []
    l_ops, g_ops = _run(h, feats, targets, replay, cfg, fused=False, pin=pin)
    _lib.profile_enable(True)
    l_fused, g_fused = _run(h, feats, targets, replay, cfg, fused=True, pin=pin)
    torch.cuda.synchronize()
    n_small, _, _ = _lib.profile_get("small_gemm_kernel")
    _lib.profile_enable(False)
    assert n_small > 0
    for k in l_ops:
        assert abs(l_ops[k] - l_fused[k]) <= 2e-3 * max(1.0, abs(l_ops[k])), (k, l_ops[k], l_fused[k])
    assert set(g_ops) == set(g_fused)
    worst = ("", 0.0)
    for n in g_ops:
        a, b = g_ops[n].double().flatten(), g_fused[n].double().flatten()
        rel = (a - b).norm().item() / (a.norm().item() + 1e-12)
        if rel > worst[1]:
            worst = (n, rel)
    # identical kernels; differences are the order of bf16 gradient accumulation and of the LayerNorm atomics
    assert worst[1] < 2e-2, worst


def test_decoder_layer_rejects_bad_struct():
    import ctypes
    from mp_former_amd import _lib
    from mp_former_amd.decoder_layer import MpfDecoderLayer
    L = MpfDecoderLayer()
    L.Qt, L.N, L.H, L.S, L.ffn_dim = 10, 1, 8, 64, 2048
    assert _lib.lib().mpf_decoder_layer_forward(ctypes.byref(L), None) == -3      # MPF_E_NULL
    L.H = 4
    assert _lib.lib().mpf_decoder_layer_forward(ctypes.byref(L), None) == -2      # MPF_E_SHAPE


@pytest.mark.parametrize("T,H,W,size", [(5, 64, 64, (8, 8)), (3, 96, 160, (12, 20)), (7, 1024, 1024, (128, 128)),
                                        (2, 1024, 1024, (32, 32)), (4, 60, 60, (20, 12)), (1, 48, 36, (16, 12))])
def test_mp_rows_block_empty_matches_area_interpolation(T, H, W, size):
    """gt_block_or == (F.interpolate(masks, size, mode='area') <= 1e-8), mask2former_transformer_decoder.py:986-987"""
    import torch.nn.functional as F
    from mp_former_amd.transformer_decoder import gt_block_or
    g = torch.Generator(device="cpu").manual_seed(T * H + W)
    masks = torch.zeros(T, H, W, dtype=torch.bool)
    for t in range(T):      # sparse blobs + isolated pixels, some maps empty
        if t % 4 == 3:
            continue
        y0, x0 = int(torch.randint(0, H // 2, (1,), generator=g)), int(torch.randint(0, W // 2, (1,), generator=g))
        masks[t, y0:y0 + H // 3, x0:x0 + W // 4] = True
        masks[t, int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))] = True
    masks = masks.cuda()
    ref = F.interpolate(masks.float().unsqueeze(1), size=size, mode="area").flatten(1) <= 1e-8
    got = gt_block_or(masks, size)
    assert got.dtype == torch.bool and torch.equal(got, ref)


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_decoder_inputs_kernel_matches_torch(out_dtype):
    """src = x + level_embed, kin = src + pos, sequence-first (mask2former_transformer_decoder.py:1756-1764), from a
    channel-last view of a larger memory tensor, forward and backward"""
    from mp_former_amd.transformer_decoder import _DecoderInputs
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    N, C, H, W = 2, 256, 12, 20
    S = H * W
    mem = torch.randn(N, 500, C, device=dev)                   # this level = rows 100 .. 100 + S of the encoder memory
    base = mem[:, 100:100 + S].detach().clone().requires_grad_(True)
    le = torch.randn(C, device=dev, requires_grad=True)
    pos = torch.randn(C, H, W, device=dev)
    pos_t = pos.flatten(1).t().contiguous()

    def view(t):
        return t.transpose(1, 2).reshape(N, C, H, W)

    x = view(base)
    assert x.stride(1) == 1
    src, kin = _DecoderInputs.apply(x, le, pos_t, out_dtype)
    g1 = torch.randn(S, N, C, device=dev).to(out_dtype)
    g2 = torch.randn(S, N, C, device=dev).to(out_dtype)
    torch.autograd.backward([src, kin], [g1, g2])
    gb, gl = base.grad.clone(), le.grad.clone()
    base.grad = None; le.grad = None
    xr = view(base)
    s_ref = (xr.flatten(2) + le[None, :, None]).permute(2, 0, 1)
    k_ref = s_ref + pos.flatten(1).t()[:, None, :]
    s_c, k_c = s_ref.to(out_dtype), k_ref.to(out_dtype)
    assert torch.equal(src, s_c) and torch.equal(kin, k_c) and src.is_contiguous() and kin.is_contiguous()
    torch.autograd.backward([s_c, k_c], [g1, g2])
    torch.testing.assert_close(gb, base.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gl, le.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("switch", ["MPF_KV_BATCH", "MPF_NATIVE_LOSS_TAIL", "MPF_GN_FLATTEN", "MPF_HEADS_TALL"])
def test_round2_switches_do_not_change_the_step(switch):
    """Every scheduling change of the second half of round 2 (key / value projections per level,
    native criterion tail, GroupNorm into the flattened encoder input, tall-linear heads) against its plainer route
    (`<switch>=0`): same losses, same gradients up to bf16 accumulation-order noise, on the AMP path with replayed draws."""
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_ragged")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    # (the assignment is pinned across the two runs, see _run: the two routes round differently, the toy costs tie)
    pin = []
    l_on, g_on = _run(h, feats, targets, replay, cfg, fused=True, pin=pin)
    os.environ[switch] = "0"
    try:
        l_off, g_off = _run(h, feats, targets, replay, cfg, fused=True, pin=pin)
    finally:
        os.environ.pop(switch, None)
    assert set(g_on) == set(g_off) and set(l_on) == set(l_off)
    print(f"{switch}: max relative loss difference {max(abs(l_on[k] - l_off[k]) / max(1.0, abs(l_on[k])) for k in l_on):.2e}")
    bad = [(k, l_on[k], l_off[k]) for k in l_on if abs(l_on[k] - l_off[k]) > 2e-3 * max(1.0, abs(l_on[k]))]
    assert not bad, (switch, bad[:3])
    worst = ("", 0.0)
    for n in g_on:
        a, b = g_on[n].double().flatten(), g_off[n].double().flatten()
        rel = (a - b).norm().item() / (a.norm().item() + 1e-12)
        if rel > worst[1]:
            worst = (n, rel)
    assert worst[1] < 2e-2, (switch, worst)


@pytest.mark.parametrize("rows", [1, 16, 230, 603])
def test_row_chain_out_projection_residual_layernorm(rows):
    """``mpf_lin256_res_ln_forward`` (csrc/row_chain.hip: output projection + residual + LayerNorm in one workgroup per 16 rows,
    mask2former_transformer_decoder.py:42-52) against plain torch fp32 on the same bf16 operands with the same rounding point (the
    projection's result in bf16), and against the two separate kernels it replaces."""
    from mp_former_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(rows)
    a = torch.randn(rows, 256, generator=g).to(dev).bfloat16()
    w = (torch.randn(256, 256, generator=g) / 16).to(dev).bfloat16()
    b = torch.randn(256, generator=g).to(dev).bfloat16()
    x = torch.randn(rows, 256, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    beta = (0.1 * torch.randn(256, generator=g)).to(dev)
    s = torch.empty(rows, 256, device=dev)
    y32 = torch.empty_like(s)
    y16 = torch.empty(rows, 256, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    lib = _lib.lib()
    st = _lib.stream_ptr(dev)
    _lib.check(lib.mpf_lin256_res_ln_forward(a.data_ptr(), w.data_ptr(), b.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             s.data_ptr(), y32.data_ptr(), y16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, 1e-5, st), "chain")
    t = (a.float() @ w.float().t() + b.float()).bfloat16().float()
    s_ref = x + t
    y_ref = torch.nn.functional.layer_norm(s_ref, (256,), gamma, beta, 1e-5)
    # the projection in fp32 order-of-summation noise may flip a bf16 rounding of t: one bf16 ulp of |t| <~ 8
    assert float((s - s_ref).abs().max()) <= 2 ** -5
    assert float(((s - s_ref).abs() > 1e-6).float().mean()) < 2e-3
    torch.testing.assert_close(y32, torch.nn.functional.layer_norm(s, (256,), gamma, beta, 1e-5), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y16.float(), y32.bfloat16().float(), rtol=0, atol=0)
    torch.testing.assert_close(mean, s.mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd, (s.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-5, atol=1e-6)
    assert float((y32 - y_ref).abs().max()) < 0.1
    # the separate kernels it replaces: bit for bit (same contraction order, same row sums)
    t2 = torch.empty(rows, 256, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.mpf_small_gemm_bf16(a.data_ptr(), 256, 1, None, w.data_ptr(), 256, 1, b.data_ptr(), None, 0, t2.data_ptr(), 256, None,
                                       rows, 256, 256, 0, st), "gemm")
    s2, y2, y2h = torch.empty_like(s), torch.empty_like(s), torch.empty_like(y16)
    m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
    _lib.check(lib.mpf_res_ln256_forward(x.data_ptr(), t2.data_ptr(), _lib.MPF_BF16, gamma.data_ptr(), beta.data_ptr(), s2.data_ptr(),
                                         y2.data_ptr(), y2h.data_ptr(), m2.data_ptr(), r2.data_ptr(), rows, 1e-5, None, 0, None, st), "ln")
    assert torch.equal(s, s2) and torch.equal(y32, y2) and torch.equal(y16, y2h) and torch.equal(mean, m2) and torch.equal(rstd, r2)


@pytest.mark.parametrize("rows", [3, 64, 219, 600])
def test_row_chain_decoder_norm_mask_embed(rows):
    """``mpf_ln256_mlp3_forward`` (decoder_norm + the three mask_embed layers in one workgroup per 16 rows,
    mask2former_transformer_decoder.py:1859-1866) against torch with bf16 rounding of every stored intermediate."""
    from mp_former_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(100 + rows)
    x = (2 * torch.randn(rows, 256, generator=g)).to(dev)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    beta = (0.1 * torch.randn(256, generator=g)).to(dev)
    ws = [(torch.randn(256, 256, generator=g) / 16).to(dev).bfloat16() for _ in range(3)]
    bs = [(0.2 * torch.randn(256, generator=g)).to(dev).bfloat16() for _ in range(3)]
    out = torch.empty(rows, 256, device=dev, dtype=torch.bfloat16)
    _lib.check(_lib.lib().mpf_ln256_mlp3_forward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(),
                                                 ws[1].data_ptr(), bs[1].data_ptr(), ws[2].data_ptr(), bs[2].data_ptr(), out.data_ptr(), rows,
                                                 1e-5, _lib.stream_ptr(dev)), "mlp3")
    e = torch.nn.functional.layer_norm(x, (256,), gamma, beta, 1e-5).bfloat16().float()
    e = torch.relu(e @ ws[0].float().t() + bs[0].float()).bfloat16().float()
    e = torch.relu(e @ ws[1].float().t() + bs[1].float()).bfloat16().float()
    e = (e @ ws[2].float().t() + bs[2].float()).bfloat16().float()
    err = (out.float() - e).abs()
    # a flipped bf16 rounding of an intermediate moves an output by a fraction of ITS bf16 ulp: a few ulps at most anywhere
    assert float(err.max()) <= 4 * 2 ** -8 * float(e.abs().max())
    assert float((err > 0).float().mean()) < 0.05
    # the four launches it replaces: bit for bit
    lib, st = _lib.lib(), _lib.stream_ptr(dev)
    d16, e1, e2 = (torch.empty(rows, 256, device=dev, dtype=torch.bfloat16) for _ in range(3))
    m2, r2 = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    _lib.check(lib.mpf_res_ln256_forward(x.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(), None, None, d16.data_ptr(), m2.data_ptr(),
                                         r2.data_ptr(), rows, 1e-5, None, 0, None, st), "ln")
    for src, dst, k, relu in ((d16, e1, 0, 1), (e1, e2, 1, 1), (e2, d16, 2, 0)):
        _lib.check(lib.mpf_small_gemm_bf16(src.data_ptr(), 256, 1, None, ws[k].data_ptr(), 256, 1, bs[k].data_ptr(), None, 0, dst.data_ptr(), 256,
                                           None, rows, 256, 256, relu, st), "gemm")
    assert torch.equal(out, d16)


def test_row_chains_do_not_change_the_step():
    """lib option ``decoder_row_chain`` (1: csrc/row_chain.hip, 0: separate projection / LayerNorm / MLP launches): same losses, same
    gradients up to bf16 noise on the AMP path with replayed draws and a pinned assignment."""
    from mp_former_amd import _lib
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_ragged")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    pin = []
    _lib.profile_enable(True)
    l_on, g_on = _run(h, feats, targets, replay, cfg, fused=True, pin=pin)
    torch.cuda.synchronize()
    n_chain = _lib.profile_get("lin256_res_ln_kernel")[0]
    _lib.profile_enable(False)
    assert n_chain > 0
    _lib.set_option("decoder_row_chain", 0)
    try:
        l_off, g_off = _run(h, feats, targets, replay, cfg, fused=True, pin=pin)
    finally:
        _lib.set_option("decoder_row_chain", 1)
    bad = [(k, l_on[k], l_off[k]) for k in l_on if abs(l_on[k] - l_off[k]) > 2e-3 * max(1.0, abs(l_on[k]))]
    assert not bad, bad[:3]
    worst = ("", 0.0)
    for n in g_on:
        a, b = g_on[n].double().flatten(), g_off[n].double().flatten()
        rel = (a - b).norm().item() / (a.norm().item() + 1e-12)
        if rel > worst[1]:
            worst = (n, rel)
    assert worst[1] < 2e-2, worst
