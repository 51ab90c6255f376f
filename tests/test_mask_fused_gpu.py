"""GPU parity of the fused mask product kernels (csrc/mask_fused.hip through the C ABI):
  * mpf_match_cost_fused against the oracle's matcher cost (oracle/head_ref.py:matcher_cost = matcher.py:105-148) evaluated
    on the MATERIALISED fp32 map einsum("qc,chw->qhw") — i.e. the reference's formulation on identical inputs;
  * mpf_pair_planes_forward / _backward and the full product against the einsum and its autograd gradients.
Inputs are bf16-representable, so the only differences are fp32 summation order, the 16-bit interpolated feature planes
(matching cost) and the bf16 rounding of the outputs (planes, gradients): tolerances are written at each assert."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _factors(N, R, H, W, dev, seed=0, scale=0.2):
    """me as the heads produce it: an [N, R, C] view of a sequence-first [R, N, C] tensor; mf channel-last planes"""
    g = torch.Generator().manual_seed(seed)
    me = (torch.randn(R, N, 256, generator=g) * scale).to(torch.bfloat16).to(dev).transpose(0, 1)
    mf = torch.randn(N, H, W, 256, generator=g).to(torch.bfloat16).to(dev).permute(0, 3, 1, 2)
    return me, mf


@pytest.mark.parametrize("N,R,H,W", [(2, 23, 16, 16), (1, 130, 64, 64), (3, 40, 8, 48)])
def test_full_product_and_gradients_match_einsum(dev, N, R, H, W):
    from mp_former_amd import _lib, mask_fused
    me, mf = _factors(N, R, H, W, dev, seed=R)
    assert mask_fused.supported(me, mf)
    a = me.detach().clone().requires_grad_(True)          # (clone keeps the strides of the transposed view)
    assert a.stride() == me.stride()
    b = mf.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
    out = mask_fused.full_product(a, b)
    assert _lib.last_kernel() == "pair_planes_fwd_kernel"
    a32 = me.detach().float().requires_grad_(True)
    b32 = mf.detach().float().requires_grad_(True)
    ref = torch.einsum("bqc,bchw->bqhw", a32, b32)
    # bf16 result of an fp32-accumulated product: half an ulp = 2^-9 relative
    torch.testing.assert_close(out.float(), ref, rtol=5e-3, atol=5e-3 * float(ref.abs().max()) / 16)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(dev)
    out.backward(g)
    ref.backward(g.float())
    assert a.grad.stride() == a.stride()
    for got, want in ((a.grad, a32.grad), (b.grad, b32.grad)):
        err = (got.float() - want).abs().max() / want.abs().max()
        assert float(err) < 6e-3, float(err)                # bf16 rounding of the stored gradient


def _pairs_case(N, R, counts, dev, seed):
    rng = np.random.default_rng(seed)
    bi = np.concatenate([np.full(c, b) for b, c in enumerate(counts)]).astype(np.int64)
    perm = rng.permutation(len(bi))
    bi = bi[perm]                                          # pair order != slot order, as in the criterion
    rows = np.concatenate([rng.permutation(R)[:c] for c in counts])[perm] if len(bi) else np.zeros(0, np.int64)
    # each image's rows distinct: a query is matched once per output
    seen = {}
    for i, (b, r) in enumerate(zip(bi, rows)):
        while (b, rows[i]) in seen:
            rows[i] = (rows[i] + 1) % R
        seen[(b, rows[i])] = 1
    return bi, rows


@pytest.mark.parametrize("N,R,H,W,counts", [(2, 60, 16, 16, (5, 37)), (3, 200, 64, 64, (131, 0, 64)), (2, 300, 256, 256, (250, 217))])
def test_pair_planes_forward_backward_on_gathered_rows(dev, N, R, H, W, counts):
    """the criterion's use: planes of (image, row) pairs listed in arbitrary order, slots image by image; config-B size last"""
    from mp_former_amd import mask_fused
    from mp_former_amd._h2d import upload
    from mp_former_amd.criterion import SetCriterion
    me, mf = _factors(N, R, H, W, dev, seed=7)
    bi, rows = _pairs_case(N, R, counts, dev, seed=3)
    n = len(bi)
    slot, first, count = SetCriterion._slot_layout(bi, N)
    inv = np.zeros(n, dtype=np.int32)
    inv[slot] = np.arange(n, dtype=np.int32)
    row_off = upload(bi * me.stride(0) + rows * me.stride(1), dev)
    i32 = upload(np.concatenate([inv, first, count]).astype(np.int32), dev)
    a = me.detach().clone().requires_grad_(True)
    b = mf.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
    planes = mask_fused.PairPlanes.apply(a, b, row_off, i32[:n], i32[n:n + N], i32[n + N:], n, int(count.max()))
    a32 = me.detach().float().requires_grad_(True)
    b32 = mf.detach().float().requires_grad_(True)
    bi_t, rows_t = torch.from_numpy(bi).to(dev), torch.from_numpy(rows).to(dev)
    slot_t = torch.from_numpy(slot).to(dev)
    ref_pairs = torch.einsum("pc,pcx->px", a32[bi_t, rows_t], b32.flatten(2)[bi_t])            # [pairs, HW] in pair order
    ref = torch.zeros_like(ref_pairs).index_copy(0, slot_t, ref_pairs)                          # slot order
    torch.testing.assert_close(planes.float(), ref, rtol=5e-3, atol=5e-3 * float(ref.abs().max()) / 16)
    g = (torch.randn(planes.shape, generator=torch.Generator().manual_seed(9)) * 0.1).to(torch.bfloat16).to(dev)
    planes.backward(g)
    ref.backward(g.float())
    for name, got, want in (("d_embed", a.grad, a32.grad), ("d_features", b.grad, b32.grad)):
        err = (got.float() - want).abs().max() / want.abs().max()
        assert float(err) < 6e-3, (name, float(err))
    # rows that are in no pair get exactly zero
    used = torch.zeros(N, R, dtype=torch.bool, device=dev)
    used[bi_t, rows_t] = True
    assert float(a.grad[~used].abs().max()) == 0.0
    # deterministic: fixed-order reductions, no atomics — forward planes AND both gradients bit-equal run to run
    g1a, g1b = a.grad.clone(), b.grad.clone()
    a.grad = None
    b.grad = None
    planes2 = mask_fused.PairPlanes.apply(a, b, row_off, i32[:n], i32[n:n + N], i32[n + N:], n, int(count.max()))
    planes2.backward(g)
    assert torch.equal(planes2, planes)
    assert torch.equal(a.grad, g1a), "d_embed differs between two runs"
    assert torch.equal(b.grad, g1b), "d_features differs between two runs"
    # an image without pairs gets an exactly-zero feature gradient
    for bimg in range(N):
        if int(count[bimg]) == 0:
            assert float(b.grad[bimg].abs().max()) == 0.0


@pytest.mark.parametrize("L,N,Q,Qt,H,W,P,counts", [(2, 2, 10, 14, 16, 16, 112, (3, 5)), (3, 2, 100, 120, 64, 64, 1000, (17, 0)),
                                                   (2, 1, 200, 230, 32, 64, 12544, (70,)), (10, 2, 100, 114, 256, 256, 12544, (5, 14))])
def test_match_cost_fused_vs_oracle_matcher_cost(dev, L, N, Q, Qt, H, W, P, counts):
    """mask + dice cost of every (output, image, query, target) from the factors == the oracle's matcher_cost on the
    materialised fp32 maps, same points, same ground-truth masks.  Last case: config B (10 outputs, 1024^2 -> 256^2 maps,
    12 544 points); third: 200 queries in two query groups, 70 targets, points outside [0, 1] included."""
    from mp_former_amd import _lib
    from mp_former_amd.mask_fused import FactoredMasks
    from mp_former_amd.matcher import HungarianMatcher
    from oracle import head_ref as O
    g = torch.Generator().manual_seed(P + Q)
    me, mf = _factors(N, L * Qt, H, W, dev, seed=1, scale=0.15)
    root = FactoredMasks(me, mf)
    views = [root[:, l * Qt:(l + 1) * Qt][:, -Q:] for l in range(L)]
    K = 7
    logits = [torch.randn(N, Q, K + 1, generator=g).to(dev) for _ in range(L)]
    targets = []
    for b, T in enumerate(counts):
        m = torch.zeros(T, 4 * H, 4 * W, dtype=torch.bool)
        for t in range(T):
            y0, x0 = (7 * t + 3 * b) % (3 * H), (11 * t + 5) % (3 * W)
            m[t, y0:y0 + H + t, x0:x0 + W // 2 + 2 * t] = True
        targets.append({"labels": torch.randint(0, K, (T,), generator=g).to(dev), "masks": m.to(dev)})
    coords = torch.rand(L * N, P, 2, generator=g) * 1.1 - 0.05                   # a few points outside the image: zero padding
    from mp_former_amd import _rng
    tags = ["match"] + [f"match_{i}" for i in range(L - 1)]
    _rng.install_replay({tags[l]: [coords[l * N + b][None] for b in range(N)] for l in range(L)})
    try:
        outs = [{"pred_logits": logits[l], "pred_masks": views[l]} for l in range(L)]
        matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=P)
        _lib.profile_enable(True)
        C = matcher.cost_matrices(outs, targets, tags=tags)
        assert _lib.profile_get("match_cost_fused_kernel")[0] == 1
        _lib.profile_enable(False)
    finally:
        _rng.install_replay(None)
    C = C.cpu()
    me32, mf32 = me.float().cpu(), mf.float().cpu()
    for l in range(L):
        for b, T in enumerate(counts):
            if T == 0:
                continue
            maps = torch.einsum("qc,chw->qhw", me32[b, l * Qt + Qt - Q:(l + 1) * Qt], mf32[b])
            want = O.matcher_cost(logits[l][b].cpu(), maps, targets[b]["labels"].cpu(), targets[b]["masks"].cpu(),
                                  coords[l * N + b][None])
            got = C[l, b, :, :T]
            # fp32 sums over P points in another order + 2^-17 features: a few 1e-5 of costs of O(1..10)
            torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    # bit-reproducible (fixed-order reduction of the per-workgroup partial sums)
    _rng.install_replay({tags[l]: [coords[l * N + b][None] for b in range(N)] for l in range(L)})
    try:
        C2 = matcher.cost_matrices(outs, targets, tags=tags).cpu()
    finally:
        _rng.install_replay(None)
    assert torch.equal(C, C2)
