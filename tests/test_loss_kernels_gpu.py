"""GPU parity of the native criterion / decoder helper kernels (csrc/loss.hip, csrc/decoder.hip)
against plain-PyTorch fp32 references of the same ops (oracle/head_ref.py functions where they
exist).  These are floating-point kernels: tolerances are written next to each assert."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _grid_sample(x, coords):
    return F.grid_sample(x, 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros",
                         align_corners=False).squeeze(3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.bool])
def test_point_sample_matches_grid_sample(dtype):
    from mp_former_amd.point_sample import point_sample_offsets
    g = torch.Generator().manual_seed(0)
    R, h, w, P = 7, 24, 40, 1000
    x = torch.randn(R, h, w, generator=g)
    x = (x > 0) if dtype == torch.bool else x.to(dtype)
    x = x.to(DEV)
    coords = (torch.rand(R, P, 2, generator=g) * 1.2 - 0.1).to(DEV)      # incl. points outside [0,1]
    coords[0, :4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.5 / w, 0.5 / h], [1.0 - 0.5 / w, 0.3]], device=DEV)
    offs = (torch.arange(R, device=DEV) * h * w).long()
    got = point_sample_offsets(x.data_ptr(), x.dtype if dtype != torch.bool else torch.uint8, h, w, offs, coords, None,
                               torch.device(DEV))
    ref = _grid_sample(x.float()[:, None], coords)[:, 0]
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    # shared coordinate rows (the matcher's use): every map row samples coords[crow[i]]
    crow = torch.tensor([0, 0, 1, 1, 2, 2, 2], dtype=torch.int32, device=DEV)
    got2 = point_sample_offsets(x.data_ptr(), x.dtype if dtype != torch.bool else torch.uint8, h, w, offs, coords[:3].contiguous(),
                                crow, torch.device(DEV))
    ref2 = _grid_sample(x.float()[:, None], coords[crow.long()])[:, 0]
    torch.testing.assert_close(got2, ref2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,M,k", [(5, 37632, 9408), (3, 1000, 250), (1, 336, 84), (4, 4096, 4096)])
def test_select_uncertain_equals_topk_set(n, M, k):
    """the k smallest |v| per row, as a SET (order inside is free; ties at the threshold by index)"""
    from mp_former_amd.point_sample import select_uncertain
    g = torch.Generator().manual_seed(1)
    vals = torch.randn(n, M, generator=g).to(DEV)
    vals[0, :10] = 0.0                                     # ties
    coords = torch.rand(n, M, 2, generator=g).to(DEV)
    coords[..., 0] = torch.arange(M, device=DEV)[None].float()     # x carries the source index
    out = select_uncertain(vals, coords, k, k + 3)
    idx = out[:, :k, 0].long()
    for r in range(n):
        got = set(idx[r].tolist())
        assert len(got) == k
        thr = vals[r].abs().kthvalue(k).values.item()
        a = vals[r].abs()
        must = set(torch.nonzero(a < thr).flatten().tolist())
        may = set(torch.nonzero(a <= thr).flatten().tolist())
        assert must <= got <= may
        # and the y coordinate travelled with it
        torch.testing.assert_close(out[r, :k, 1], coords[r, idx[r], 1])


@pytest.mark.parametrize("hw,M,k", [((64, 64), 3000, 750), ((256, 256), 37632, 9408), ((32, 48), 1024, 1024), ((16, 16), 700, 1)])
def test_sample_select_uncertain_equals_two_step(hw, M, k):
    """fused LDS kernel (sample the plane + radix select in registers) == point_sample followed by
    select_uncertain: same coordinates in the same order."""
    from mp_former_amd import _lib
    from mp_former_amd.point_sample import MapSet, point_sample_offsets, sample_select_uncertain, select_uncertain
    g = torch.Generator().manual_seed(7)
    h, w = hw
    maps = (torch.randn(2, 5, h, w, generator=g) * 2).to(torch.bfloat16).to(DEV)
    maps[0, 1].round_()                       # plenty of ties in one plane
    ms = MapSet([maps])
    ti = np.zeros(4, np.int64); bi = np.array([0, 0, 1, 1]); qi = np.array([1, 4, 0, 2])
    offs = torch.from_numpy(ms.offsets(ti, bi, qi)).to(DEV)
    coords = torch.rand(4, M, 2, generator=g).to(DEV)
    P_out = k + 5
    got = sample_select_uncertain(ms, offs, coords, k, P_out)
    assert "sample_select_kernel" in _lib.last_kernel(), _lib.last_kernel()
    logits = point_sample_offsets(ms.base_ptr, ms.dtype, h, w, offs, coords, None, torch.device(DEV))
    ref = select_uncertain(logits, coords, k, P_out)
    assert torch.equal(got[:, :k], ref[:, :k])


@pytest.mark.parametrize("Q,dtype,group", [(9, torch.float32, 1), (12, torch.bfloat16, 12), (12, torch.bfloat16, 1)])
def test_match_cost_matches_oracle(Q, dtype, group):
    """mask + dice part of the matching cost (matcher.py:15-62,122-148) vs oracle.matcher_cost; bf16 maps
    with a row-group guarantee take the LDS-staged kernel (4 rows per workgroup)."""
    from oracle import head_ref as O
    from mp_former_amd import _lib
    from mp_former_amd.point_sample import MapSet, match_cost, point_sample_offsets
    g = torch.Generator().manual_seed(2)
    N, h, w, H, W, P = 2, 16, 16, 64, 64, 500
    T = [3, 11]
    masks = (torch.randn(N, Q, h, w, generator=g) * 3).to(dtype)
    gts = [(torch.rand(t, H, W, generator=g) < 0.3) for t in T]
    coords = torch.rand(N, P, 2, generator=g)
    ms = MapSet([masks.to(DEV)])
    gt_u8 = torch.cat(gts).to(DEV).view(torch.uint8)
    Tt, Tmax = sum(T), max(T)
    b = np.repeat(np.arange(N), Q); q = np.tile(np.arange(Q), N)
    offs = torch.from_numpy(ms.offsets(np.zeros(N * Q, np.int64), b, q)).to(DEV)
    gt_offs = (torch.arange(Tt, device=DEV) * H * W).long()
    img_of = torch.tensor([0] * T[0] + [1] * T[1], dtype=torch.int32, device=DEV)
    tsamp = point_sample_offsets(gt_u8.data_ptr(), torch.uint8, H, W, gt_offs, coords.to(DEV), img_of, torch.device(DEV))
    first = torch.tensor([0] * Q + [T[0]] * Q, dtype=torch.int32, device=DEV)
    cnt = torch.tensor([T[0]] * Q + [T[1]] * Q, dtype=torch.int32, device=DEV)
    C = match_cost(ms, offs, coords.to(DEV), torch.from_numpy(b.astype(np.int32)).to(DEV), tsamp, first, cnt, Tmax, 5.0, 5.0,
                   rows_per_group=group)
    assert ("match_cost_lds" in _lib.last_kernel()) == (group > 1 and dtype == torch.bfloat16), _lib.last_kernel()
    C = C.view(N, Q, Tmax).cpu()
    for i in range(N):
        logits = torch.zeros(Q, 4)          # class cost cancels: use zero weight
        ref = O.matcher_cost(logits, masks[i].float(), torch.zeros(T[i], dtype=torch.long), gts[i], coords[i:i + 1],
                             w_class=0.0, w_mask=5.0, w_dice=5.0)
        torch.testing.assert_close(C[i, :, :T[i]], ref, rtol=2e-4, atol=2e-4)


def test_bit_packed_masks_sample_like_byte_masks():
    """mpf_pack_mask_bits + the MPF_BITS source of mpf_point_sample == sampling the byte masks."""
    from mp_former_amd.matcher import GTMasks
    from mp_former_amd.point_sample import point_sample_offsets
    g = torch.Generator().manual_seed(11)
    H, W, P = 96, 64, 3000
    masks = (torch.rand(7, H, W, generator=g) < 0.35).to(DEV)
    gtm = GTMasks([{"masks": masks[:3]}, {"masks": masks[3:]}])
    assert gtm.bits is not None
    coords = torch.rand(2, P, 2, generator=g).to(DEV) * 1.1 - 0.05          # some points outside
    offs = (torch.arange(7, device=DEV) * H * W).long()
    rows = torch.tensor([0, 0, 0, 1, 1, 1, 1], dtype=torch.int32, device=DEV)
    a = point_sample_offsets(gtm.u8.data_ptr(), torch.uint8, H, W, offs, coords, rows, torch.device(DEV))
    b = point_sample_offsets(gtm.bits.data_ptr(), "bits", H, W, offs, coords, rows, torch.device(DEV))
    assert torch.equal(a, b)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("hw", [(32, 32), (200, 176)])        # one LDS band / two bands in the backward
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mask_loss_forward_backward_matches_autograd(dtype, hw, packed):
    """fused BCE + dice sums and their gradient vs autograd through grid_sample + the reference losses
    (criterion.py:21-65,172-191).  bf16 maps are compared on the bf16-rounded values."""
    from mp_former_amd.point_sample import MapSet, MaskLossSums
    g = torch.Generator().manual_seed(3)
    N, Q, (h, w), H, W, P = 2, 6, hw, 128, 128, 784
    a = (torch.randn(N, Q, h, w, generator=g) * 2).to(dtype).to(DEV).requires_grad_(True)
    bten = (torch.randn(N, 3, h, w, generator=g) * 2).to(dtype).to(DEV).requires_grad_(True)
    gt = (torch.rand(5, H, W, generator=g) < 0.4).to(DEV)
    ms = MapSet([a, bten])
    ti = np.array([0, 0, 1, 0, 1]); bi = np.array([0, 1, 1, 1, 0]); qi = np.array([2, 5, 1, 0, 2]); gr = np.array([0, 3, 4, 1, 2])
    n = len(ti)
    coords = torch.rand(n, P, 2, generator=g).to(DEV)
    po = torch.from_numpy(ms.offsets(ti, bi, qi)).to(DEV)
    go = torch.from_numpy(ms.grad_offsets(ti, bi, qi)).to(DEV)
    gt_rows = torch.from_numpy(gr.astype(np.int32)).to(DEV)
    if packed:
        from mp_former_amd.matcher import GTMasks
        gsrc = GTMasks([{"masks": gt}])
        assert gsrc.bits is not None
    else:
        gsrc = gt.view(torch.uint8)
    sums = MaskLossSums.apply(ms, po, go, gsrc, gt_rows, coords, a, bten)
    wts = torch.tensor([[1.0, -2.0, 0.5, 0.0]], device=DEV) * torch.arange(1, n + 1, device=DEV)[:, None]
    (sums * wts).sum().backward()
    ga, gb = a.grad.float().clone(), bten.grad.float().clone()
    # reference
    a2 = a.detach().float().requires_grad_(True); b2 = bten.detach().float().requires_grad_(True)
    maps = [a2, b2]
    rows = torch.stack([maps[t][b_, q_] for t, b_, q_ in zip(ti, bi, qi)])[:, None]
    x = _grid_sample(rows, coords)[:, 0]
    t = _grid_sample(gt[gr].float()[:, None], coords)[:, 0]
    s = x.sigmoid()
    ref = torch.stack([F.binary_cross_entropy_with_logits(x, t, reduction="none").sum(1), (s * t).sum(1), s.sum(1), t.sum(1)], 1)
    torch.testing.assert_close(sums.detach(), ref, rtol=2e-4, atol=2e-3)
    (ref * wts).sum().backward()
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4      # returned grad is rounded to the map dtype
    torch.testing.assert_close(ga, a2.grad, rtol=tol, atol=tol)
    torch.testing.assert_close(gb, b2.grad, rtol=tol, atol=tol)


@pytest.mark.parametrize("size,hw", [((8, 8), (64, 64)), ((16, 16), (64, 64)), ((32, 32), (64, 64)), ((5, 7), (40, 56))])
def test_attn_mask_matches_reference_formula(size, hw):
    """decoder :1869-1875 + :1814-1816 + :1780 vs the same in torch ops."""
    from mp_former_amd.transformer_decoder import native_attn_mask
    g = torch.Generator().manual_seed(4)
    N, Q, pad = 2, 11, 3
    for dtype in (torch.float32, torch.bfloat16):
        m = (torch.randn(N, Q, *hw, generator=g) * 2 - 0.5).to(dtype).to(DEV)
        m[0, 5] = 5.0            # nothing masked
        m[1, 6] = -5.0           # everything masked -> row becomes all False
        rows = (torch.rand(N, pad, size[0] * size[1], generator=g) < 0.5).to(DEV)
        rows[0, 1] = True        # an MP padding row: all True -> all False
        got = native_attn_mask(m, size, rows)
        ref = F.interpolate(m.float(), size=size, mode="bilinear", align_corners=False).flatten(2) < 0
        ref = torch.cat([rows, ref[:, pad:]], 1)
        ref = ref & ~ref.all(-1, keepdim=True)
        # a logit within rounding of 0 may flip: allow only such positions to differ
        diff = got != ref
        if diff.any():
            v = F.interpolate(m.float(), size=size, mode="bilinear", align_corners=False).flatten(2)
            assert v[diff].abs().max() < 1e-5
        assert not got[1, 6].any() and not got[0, 1].any() and not got[0, 5].any()
        got2 = native_attn_mask(m, size, None)
        assert torch.equal(got2[:, pad:][~diff[:, pad:]], ref[:, pad:][~diff[:, pad:]])


def test_class_loss_kernel_matches_cross_entropy():
    """mpf_class_loss_forward / backward (csrc/criterion_tail.hip) == F.cross_entropy(logits^T, target, weight) per output,
    values and gradients, for strided bf16 / fp32 logits and per-output / shared targets (criterion.py:123-139)."""
    import torch.nn.functional as F
    from mp_former_amd.criterion import _ClassLossFn
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    for dtype, L, N, Q, C, shared in [(torch.float32, 10, 2, 100, 81, False), (torch.bfloat16, 10, 2, 14, 81, True),
                                      (torch.bfloat16, 3, 1, 7, 134, False), (torch.float32, 1, 3, 5, 2, True)]:
        parent = (torch.randn(N, L * Q + 3, C, device=dev) * 3).to(dtype).requires_grad_(True)
        logits = parent.as_strided((L, N, Q, C), (Q * C, (L * Q + 3) * C, C, 1), 0)        # output l = rows [l*Q, (l+1)*Q)
        tgt = torch.randint(0, C, (N, Q) if shared else (L, N, Q), device=dev)
        w = torch.ones(C, device=dev); w[-1] = 0.1
        ce = _ClassLossFn.apply(logits, tgt, w)
        g = torch.rand(L, device=dev) + 0.5
        ce.backward(g)
        got = parent.grad.clone(); parent.grad = None
        ref_in = parent.detach().double().requires_grad_(True)
        rl = ref_in.as_strided((L, N, Q, C), (Q * C, (L * Q + 3) * C, C, 1), 0)
        t3 = tgt if tgt.dim() == 3 else tgt[None].expand(L, -1, -1)
        ref = torch.stack([F.cross_entropy(rl[l].transpose(1, 2), t3[l], w.double()) for l in range(L)])
        ref.backward(g.double())
        assert torch.allclose(ce.double(), ref, rtol=2e-6, atol=1e-6), (ce, ref)
        tol = 1e-6 if dtype == torch.float32 else 4e-3
        assert (got.double() - ref_in.grad).abs().max() <= tol * ref_in.grad.abs().max() + 1e-9


def test_class_loss_out_of_range_label_is_loud_not_out_of_bounds():
    """A label outside [0, C) (ignore_index-style -100, or a dataset with more classes than the head): F.cross_entropy
    device-asserts; the native kernel reads nothing out of bounds and returns NaN for that output only."""
    from mp_former_amd.criterion import _ClassLossFn
    dev = torch.device("cuda:0")
    L, N, Q, C = 3, 2, 9, 81
    logits = torch.randn(L, N, Q, C, device=dev, requires_grad=True)
    tgt = torch.randint(0, C, (L, N, Q), device=dev)
    tgt[1, 0, 4] = C + 5
    tgt[2, 1, 0] = -100
    w = torch.ones(C, device=dev)
    ce = _ClassLossFn.apply(logits, tgt, w)
    assert bool(torch.isfinite(ce[0])) and bool(torch.isnan(ce[1])) and bool(torch.isnan(ce[2]))
    ce[0].backward()
    assert bool(torch.isfinite(logits.grad[0]).all())


def test_mask_loss_finalize_kernel_matches_the_tensor_expression():
    from mp_former_amd.criterion import _MaskLossFinalizeFn
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    G, P = 7, 12544
    counts = [5, 0, 130, 1, 64, 65, 9]
    order = [0, 4, 1, 5, 2, 6, 3]                  # groups laid out in a different order than their index
    runs = torch.zeros(G, 2, dtype=torch.int64)
    gid, pos = [], 0
    for g in order:
        runs[g, 0], runs[g, 1] = pos, counts[g]
        gid += [g] * counts[g]
        pos += counts[g]
    n = pos
    sums = (torch.rand(n, 4, device=dev) * 50).requires_grad_(True)
    norm = torch.rand(G, device=dev) + 1
    out = _MaskLossFinalizeFn.apply(sums, runs.to(dev), norm, P)
    go = torch.rand(2, G, device=dev)
    out.backward(go)
    got = sums.grad.clone(); sums.grad = None
    s = sums.detach().double().requires_grad_(True)
    gi = torch.tensor(gid, device=dev)
    z = torch.zeros(G, dtype=torch.float64, device=dev)
    rm = z.index_add(0, gi, s[:, 0] / P) / norm.double()
    rd = z.index_add(0, gi, 1 - (2 * s[:, 1] + 1) / (s[:, 2] + s[:, 3] + 1)) / norm.double()
    (rm * go[0].double()).sum().backward(retain_graph=True)
    (rd * go[1].double()).sum().backward()
    assert torch.allclose(out[0].double(), rm, rtol=1e-5, atol=1e-7) and torch.allclose(out[1].double(), rd, rtol=1e-5, atol=1e-7)
    assert torch.allclose(got[:, :3].double(), s.grad[:, :3], rtol=1e-5, atol=1e-9)
