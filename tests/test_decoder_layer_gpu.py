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
