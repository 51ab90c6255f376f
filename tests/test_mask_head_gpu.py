"""Fused mask head (csrc/mask_head.hip) against the reference formula of forward_prediction_heads
(mask2former_transformer_decoder.py:1869-1875): einsum("bqc,bchw->bqhw") -> F.interpolate(bilinear, align_corners=False)
-> sigmoid < 0.5, plus the mask-piloted row overwrite (:1814-1816) and the all-masked-row rule (:1780)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(me, mf, size, mp_rows):
    # me [Q, N, C] bf16, mf [N, C, h, w] bf16 -> bool [N, Q, hl*wl]; float64 accumulate so that only genuine near-zero
    # logits can disagree with the bf16-operand / fp32-accumulate product
    m = torch.einsum("qnc,nchw->nqhw", me.double(), mf.double())
    r = F.interpolate(m, size=size, mode="bilinear", align_corners=False)
    am = (r.sigmoid() < 0.5).flatten(2)
    if mp_rows is not None:
        am[:, :mp_rows.shape[1]] = mp_rows
    am[torch.where(am.sum(-1) == am.shape[-1])] = False
    return am, r.flatten(2)


@pytest.mark.parametrize("N,Q,h,w,size,pad", [(2, 117, 64, 64, (32, 32), 17), (1, 100, 64, 96, (16, 24), 0), (2, 213, 40, 40, (20, 20), 13),
                                                (1, 30, 32, 32, (8, 8), 3)])
def test_fused_mask_head_matches_reference_formula(N, Q, h, w, size, pad):
    from mp_former_amd import transformer_decoder as TD
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Q)
    me = (torch.randn(Q, N, 256, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    mf = torch.randn(N, 256, h, w, generator=g).to(torch.bfloat16).to(dev)
    HW = size[0] * size[1]
    mp_rows = None
    if pad:
        mp_rows = (torch.rand(N, pad, HW, generator=g) < 0.6).to(dev)
        mp_rows[:, 0] = True                       # an all-masked MP row: must come out all False
    me[5] = me[5].abs() * 0 + 1.0                  # a row with constant positive embedding ...
    mf_row = mf.float().sum(1, keepdim=True)
    pooled = TD.pool_features(mf, size)
    # pooled features vs F.interpolate on the bf16 features (fp32 math), rounded to bf16
    want_p = F.interpolate(mf.float(), size=size, mode="bilinear", align_corners=False).flatten(2).transpose(1, 2)
    torch.testing.assert_close(pooled.float(), want_p.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-2)
    got = TD.mask_head_bits(me, pooled, mp_rows)
    want, logits = _reference(me, mf, size, mp_rows)
    assert got.dtype == torch.bool and got.shape == want.shape
    diff = got != want
    # disagreements are allowed only where the logit is within rounding of zero (bf16 pooled operand: 2^-8 relative of
    # the terms' magnitude); rows touched by the MP overwrite / all-masked rule must agree exactly
    scale = torch.einsum("qnc,nchw->nqhw", me.double().abs(), mf.double().abs())
    scale = F.interpolate(scale, size=size, mode="bilinear", align_corners=False).flatten(2)
    near = logits.abs() <= 2.0 ** -7 * scale
    if pad:
        assert not diff[:, :pad].any()
    bad = diff & ~near
    # a row flipped by the all-masked rule because of a near-zero logit would differ everywhere: exclude rows whose
    # open pixels are all near-zero
    rows_fragile = ((~want) & ~near).sum(-1) == 0
    bad &= ~rows_fragile[..., None]
    assert int(bad.sum()) == 0, f"{int(bad.sum())} mask bits differ away from zero logits (of {diff.numel()}; {int(diff.sum())} near zero)"
    assert float(diff.float().mean()) < 5e-3


def test_all_masked_row_is_cleared_and_flags_reset():
    from mp_former_amd import transformer_decoder as TD
    dev = torch.device("cuda:0")
    N, Q, HW = 2, 20, 256
    pooled = torch.ones(N, HW, 256, dtype=torch.bfloat16, device=dev)
    me = torch.ones(Q, N, 256, dtype=torch.bfloat16, device=dev)
    me[3] = -1.0                                    # logits < 0 everywhere: fully masked -> attends everywhere
    for _ in range(2):                              # second call: the scratch flags must have been reset
        got = TD.mask_head_bits(me, pooled, None)
        assert not got.any()
    me[3, 0, :] = 1.0
    me[4, 1, :] = -1.0
    pooled[1, :7] = -1.0                            # image 1: pixels 0..6 flip sign
    got = TD.mask_head_bits(me, pooled, None)
    assert not got[0].any()
    assert got[1, 0, :7].all() and not got[1, 0, 7:].any()          # positive embedding, negative features
    assert (~got[1, 4, :7]).all() and got[1, 4, 7:].all()           # negative embedding


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,h,w,size", [(2, 64, 64, (32, 32)), (1, 40, 56, (10, 14)), (2, 24, 24, (24, 24))])
def test_pool_features_channel_last_equals_nchw(dtype, N, h, w, size):
    """The channel-last resize (mpf_pool_features_cl) evaluates the same expression as the NCHW one: bit-equal output."""
    from mp_former_amd import _lib, transformer_decoder as TD
    dev = torch.device("cuda:0")
    torch.manual_seed(h + w)
    mf = torch.randn(N, 256, h, w, device=dev).to(dtype)
    a = TD.pool_features(mf, size)
    assert _lib.last_kernel() == "pool_features_kernel"
    b = TD.pool_features(mf.contiguous(memory_format=torch.channels_last) if N > 1 else
                         mf.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2), size)
    assert _lib.last_kernel() == "pool_features_cl_kernel"
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_mask_product_channel_last_matches_einsum(dtype):
    """mask_product on channel-last features = einsum("bqc,bchw->bqhw"), values and both gradients; bf16 operands run on
    the native MFMA product (csrc/mask_fused.hip) and the feature gradient comes back as channel-last planes; fp32 is the
    library einsum (not on the AMP training path)."""
    from mp_former_amd import _lib, transformer_decoder as TD
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    N, Q, C, H, W = 2, 77, 256, 24, 48
    me = torch.randn(N, Q, C, device=dev).to(dtype).requires_grad_(True)
    mf = torch.randn(N, H, W, C, device=dev).to(dtype).permute(0, 3, 1, 2).requires_grad_(True)
    assert TD._is_planes(mf) and not mf.is_contiguous()
    out = TD.mask_product(me, mf)
    if dtype == torch.bfloat16:
        assert _lib.last_kernel() == "pair_planes_fwd_kernel"
    ref = torch.einsum("bqc,bchw->bqhw", me.double(), mf.double())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    torch.testing.assert_close(out.double(), ref, rtol=tol, atol=tol * 16)
    g = torch.randn_like(out)
    g_me, g_mf = torch.autograd.grad(out, [me, mf], g)
    r_me, r_mf = torch.autograd.grad(ref, [me, mf], g.double())
    assert dtype != torch.bfloat16 or TD._is_planes(g_mf)
    torch.testing.assert_close(g_me.double(), r_me.double(), rtol=tol, atol=tol * float(r_me.abs().max()))
    torch.testing.assert_close(g_mf.double(), r_mf.double(), rtol=tol, atol=tol * float(r_mf.abs().max()))


@pytest.mark.parametrize("N,Q,HW,pad", [(2, 120, 1024, 20), (1, 100, 4096, 0), (2, 300, 256, 37), (3, 17, 16384, 5)])
def test_next_attn_mask_as_one_native_call_equals_the_five_ops(N, Q, HW, pad):
    """``mpf_next_attn_mask`` (decoder_norm -> mask_embed MLP -> fused mask head, mask2former_transformer_decoder.py:1859-1875
    detached) issues the same five launches as res_ln + 3 x linear + mask_head_bits: the boolean masks must be identical, incl.
    the MP rows and the all-masked-row rule, and a second call must see clean flags."""
    from mp_former_amd.resln import res_ln
    from mp_former_amd.transformer_decoder import linear, mask_head_bits, next_attn_mask_native
    dev = torch.device("cuda:0")
    torch.manual_seed(N * 1000 + Q)
    x = torch.randn(Q, N, 256, device=dev) * 2
    norm = torch.nn.LayerNorm(256).to(dev)
    with torch.no_grad():
        norm.weight.add_(torch.randn(256, device=dev) * 0.2)
        norm.bias.add_(torch.randn(256, device=dev) * 0.2)
    mlp = [((torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16), (torch.randn(256, device=dev) * 0.1).to(torch.bfloat16))
           for _ in range(3)]
    pooled = torch.randn(N, HW, 256, device=dev).to(torch.bfloat16)
    mp_rows = (torch.rand(N, pad, HW, device=dev) < 0.5) if pad else None
    if pad:
        mp_rows[:, 0] = True                 # an MP row that masks everything: the all-masked-row rule clears it
    with torch.no_grad():
        _, d16 = res_ln(norm, x, None, want32=False, want16=True)
        e = linear(d16, mlp[0][0], mlp[0][1], relu=True)
        e = linear(e, mlp[1][0], mlp[1][1], relu=True)
        me = linear(e, mlp[2][0], mlp[2][1])
        want = mask_head_bits(me, pooled, mp_rows)
        for _ in range(2):
            got = next_attn_mask_native(x, norm, mlp, pooled, mp_rows)
            assert got.dtype == torch.bool and got.shape == (N, Q, HW)
            assert torch.equal(got, want)
    assert 0.05 < float(want.float().mean()) < 0.95
