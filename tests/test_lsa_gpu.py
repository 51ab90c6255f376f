"""Device linear sum assignment (csrc/lsa.hip) against scipy.optimize.linear_sum_assignment — the call
the reference matcher makes (matcher.py:149-151): identical assignments, ties included."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


def _make(rng, mode, nr, nc):
    if mode == 0:
        return rng.standard_normal((nr, nc)).astype(np.float32)
    if mode == 1:
        return rng.integers(0, 3, (nr, nc)).astype(np.float32)          # tie-heavy
    if mode == 2:
        return np.full((nr, nc), 0.25, np.float32)                      # constant: SciPy returns the identity
    if mode == 3:
        return (rng.integers(0, 2, (nr, nc)) * 0.5 + rng.integers(0, 2, (nr, 1))).astype(np.float32)
    return (rng.standard_normal((nr, nc)) * 3 - rng.random((nr, nc))).astype(np.float32)


def _solve_many(mats, stride_pad=0):
    from mp_former_amd.lsa import FIELDS, lsa_assign
    dev = torch.device("cuda:0")
    flat, probs, pos, off = [], [], 0, 0
    for m in mats:
        nr, nc = m.shape
        padded = np.zeros((nr, nc + stride_pad), np.float32) + 99.0
        padded[:, :nc] = m
        flat.append(padded.reshape(-1))
        probs.append([off, nr, nc, nc + stride_pad, pos, 0, 7, 3, 0, 0, 0])
        pos += min(nr, nc)
        off += padded.size
    cost = torch.from_numpy(np.concatenate(flat)).to(dev)
    out = lsa_assign(cost, np.asarray(probs, np.int64), pos, want_a=True)
    rows, cols, a = out["rows"].cpu().numpy(), out["cols"].cpu().numpy(), out["a"].cpu().numpy()
    res, pos = [], 0
    for m in mats:
        k = min(m.shape)
        res.append((rows[pos:pos + k], cols[pos:pos + k], a[pos:pos + k]))
        pos += k
    return res


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lsa_matches_scipy_small(seed):
    rng = np.random.default_rng(seed)
    mats = []
    for t in range(600):
        nr, nc = int(rng.integers(1, 20)), int(rng.integers(1, 20))
        mats.append(_make(rng, t % 5, nr, nc))
    got = _solve_many(mats, stride_pad=seed)
    for m, (r, c, a) in zip(mats, got):
        er, ec = linear_sum_assignment(m)
        assert np.array_equal(r, er) and np.array_equal(c, ec), (m.shape, r, c, er, ec)
        assert np.array_equal(a, 7 + 3 * er)


def test_lsa_matches_scipy_matcher_shapes():
    """the shapes of the matcher: 100 / 200 / 300 queries x up to ~100 targets, and more targets than queries"""
    rng = np.random.default_rng(5)
    mats = []
    for (q, t) in ((100, 1), (100, 7), (100, 23), (100, 100), (200, 60), (300, 41), (100, 130), (40, 300), (300, 300)):
        for mode in (0, 1, 4):
            mats.append(_make(rng, mode, q, t))
    got = _solve_many(mats)
    for m, (r, c, _) in zip(mats, got):
        er, ec = linear_sum_assignment(m)
        assert np.array_equal(r, er) and np.array_equal(c, ec), m.shape


def test_lsa_scatter_and_bases():
    from mp_former_amd.lsa import lsa_assign
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    m = rng.standard_normal((10, 4)).astype(np.float32)
    cost = torch.from_numpy(m).to(dev)
    labels = torch.tensor([50, 51, 52, 53, 60, 61, 62, 63], dtype=torch.int64, device=dev)
    dst = torch.full((30,), -1, dtype=torch.int64, device=dev)
    out = lsa_assign(cost, np.asarray([[0, 10, 4, 4, 2, 4, 1000, 10, 5, 1, 20]], np.int64), 6, want_a=True, want_b=True,
                     scatter_dst=dst, scatter_src=labels)
    er, ec = linear_sum_assignment(m)
    assert np.array_equal(out["rows"].cpu().numpy()[2:], er)
    assert np.array_equal(out["cols"].cpu().numpy()[2:], 4 + ec)
    assert np.array_equal(out["a"].cpu().numpy()[2:], 1000 + 10 * er)
    assert np.array_equal(out["b"].cpu().numpy()[2:], 5 + er)
    exp = np.full(30, -1)
    exp[20 + er] = 60 + ec
    assert np.array_equal(dst.cpu().numpy(), exp)


def test_lsa_rejects_oversize():
    from mp_former_amd.lsa import lsa_assign
    cost = torch.zeros(600 * 2, device="cuda:0")
    with pytest.raises(RuntimeError):
        lsa_assign(cost, np.asarray([[0, 600, 2, 2, 0, 0, 0, 0, 0, 0, 0]], np.int64), 2)


def test_criterion_device_assignment_equals_host_route():
    """the whole head with the device solver (default) and with the reference's route (cost matrices to
    the host + SciPy): same pairs -> same losses and gradients"""
    import os
    from conftest import fifo_to_tags, load_head_fixture
    from test_head_gpu import _build
    from mp_former_amd import _lib, _rng
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_small")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]

    def run(device_lsa):
        os.environ["MPF_DEVICE_LSA"] = "1" if device_lsa else "0"
        h.zero_grad(set_to_none=True)
        _rng.install_replay(fifo_to_tags(replay, cfg, True))
        try:
            losses, _ = h(feats, targets)
            sum(losses.values()).backward()
        finally:
            _rng.install_replay(None)
            os.environ.pop("MPF_DEVICE_LSA", None)
        return ({k: float(v.detach()) for k, v in losses.items()},
                {n: p.grad.detach().clone() for n, p in h.named_parameters() if p.grad is not None})

    l_host, g_host = run(False)
    _lib.profile_enable(True)
    l_dev, g_dev = run(True)
    torch.cuda.synchronize()
    n_lsa, _, _ = _lib.profile_get("lsa_kernel")
    _lib.profile_enable(False)
    assert n_lsa == 1
    for k in l_host:
        assert abs(l_host[k] - l_dev[k]) <= 1e-5 * max(1.0, abs(l_host[k])), (k, l_host[k], l_dev[k])
    for n in g_host:
        a, b = g_host[n].double(), g_dev[n].double()
        assert (a - b).norm().item() <= 1e-4 * (a.norm().item() + 1e-12), n


def test_criterion_device_num_masks_path(monkeypatch):
    """multi-GPU form of the loss normaliser (num_masks kept on the device, no .item()): same losses"""
    from conftest import fifo_to_tags, load_head_fixture
    from test_head_gpu import _build
    from mp_former_amd import _rng
    import mp_former_amd.criterion as crit
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture("head_small")
    h = _build(cfg, pp, dp, dev)
    feats = {k: v.to(dev) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]

    def run():
        _rng.install_replay(fifo_to_tags(replay, cfg, True))
        try:
            losses, _ = h(feats, targets)
        finally:
            _rng.install_replay(None)
        return {k: float(v.detach()) for k, v in losses.items()}

    ref = run()
    monkeypatch.setattr(crit, "distributed", lambda: True)       # world size stays 1: same value, tensor route
    got = run()
    for k in ref:
        assert abs(ref[k] - got[k]) <= 1e-5 * max(1.0, abs(ref[k])), (k, ref[k], got[k])


@pytest.mark.parametrize("counts", [(12, 0), (0, 0), (1, 25), (10, 10)])
def test_criterion_device_assignment_edge_counts(counts):
    """no ground truth in an image / in the batch, more targets than queries (T > Q: every query matched), T == Q:
    device solver == host SciPy route, with fresh (seeded) draws"""
    import os
    from mp_former_amd.head import MPFormerHead
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    h = MPFormerHead(num_classes=7, num_queries=10, enc_layers=1, dec_layers=2, num_points=112).to(dev).train()
    size = 64
    feats = {k: torch.randn(2, c, size // s, size // s, device=dev) for k, (c, s) in
             {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}.items()}
    g = torch.Generator().manual_seed(1)
    targets = []
    for T in counts:
        masks = torch.zeros(T, size, size, dtype=torch.bool)
        for t in range(T):
            y0, x0 = int(torch.randint(0, size - 8, (1,), generator=g)), int(torch.randint(0, size - 8, (1,), generator=g))
            masks[t, y0:y0 + 4 + t % 5, x0:x0 + 3 + t % 7] = True
        targets.append({"labels": torch.randint(0, 7, (T,), generator=g).to(dev), "masks": masks.to(dev),
                        "boxes": torch.zeros(T, 4, device=dev)})

    def run(device_lsa):
        os.environ["MPF_DEVICE_LSA"] = "1" if device_lsa else "0"
        torch.manual_seed(123)
        h.zero_grad(set_to_none=True)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                losses, _ = h(feats, targets)
            total = sum(losses.values())
            total.backward()
        finally:
            os.environ.pop("MPF_DEVICE_LSA", None)
        grads = torch.cat([p.grad.flatten().float() for p in h.parameters() if p.grad is not None])
        return {k: float(v.detach()) for k, v in losses.items()}, grads

    # the forward is bit-reproducible, so both routes see the same cost matrices: one comparison decides
    l_host, g_host = run(False)
    l_dev, g_dev = run(True)
    assert set(l_host) == set(l_dev) and len(l_dev) == 6 * 3
    assert torch.isfinite(g_dev).all()
    bad = [(k, l_host[k], l_dev[k]) for k in l_host if abs(l_host[k] - l_dev[k]) > 2e-3 * max(1.0, abs(l_host[k]))]
    rel = (g_host - g_dev).norm().item() / (g_host.norm().item() + 1e-9)
    assert not bad and rel <= 2e-2, (bad[:3], rel)


def test_infeasible_cost_matrix_is_reported_like_scipy():
    """A NaN / all-infinite cost matrix: SciPy raises ValueError (matcher.py:151); the device solver sets its status
    word, which the next call (or check_status) turns into the same exception — no silent training on unassigned pairs."""
    import scipy.optimize
    from mp_former_amd import lsa as L
    dev = torch.device("cuda:0")
    L.check_status(dev, block=True)
    good = torch.rand(6, 4, device=dev)
    bad = good.clone()
    bad[2, :] = float("nan")
    with pytest.raises(ValueError):
        scipy.optimize.linear_sum_assignment(bad.cpu().numpy())
    prob = np.array([[0, 6, 4, 4, 0, 0, 0, 0, 0, 0, 0]], dtype=np.int64)
    L.lsa_assign(good.contiguous(), prob, 4)
    L.check_status(dev, block=True)                       # feasible: nothing raised
    L.lsa_assign(bad.contiguous(), prob, 4)
    with pytest.raises(ValueError, match="invalid numeric entries|infeasible"):
        L.check_status(dev, block=True)
    L.lsa_assign(good.contiguous(), prob, 4)              # the flag was cleared
    L.check_status(dev, block=True)
