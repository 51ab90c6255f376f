"""ClipAdamW (native full-model clip + AdamW, csrc/elementwise.hip) against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW — the pair the reference's FullModelGradientClippingOptimizer runs (train_net.py:316-320)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 7, 7), (256,), (300, 256), (1,), (2049,), (128, 128, 3, 3), (5, 7)]
    return [torch.randn(s, generator=g).to(dev) for s in shapes]


@pytest.mark.parametrize("max_norm", [0.01, 1e6])
def test_clip_adamw_matches_torch(max_norm):
    from mp_former_amd.optim import ClipAdamW
    dev = torch.device("cuda:0")
    init = _make(0, dev)
    init[5] = init[5].contiguous(memory_format=torch.channels_last)       # a channels-last conv weight
    pa = [t.clone().requires_grad_(True) for t in init]
    pb = [t.clone().requires_grad_(True) for t in init]
    groups = lambda ps: [{"params": ps[:3], "lr": 1e-2, "weight_decay": 0.05}, {"params": ps[3:], "lr": 1e-3, "weight_decay": 0.0}]  # noqa: E731
    oa = ClipAdamW(groups(pa), lr=1e-2, max_norm=max_norm)
    ob = torch.optim.AdamW(groups(pb), lr=1e-2)
    for it in range(4):
        grads = _make(10 + it, dev)
        for p, q, g in zip(pa, pb, grads):
            p.grad = g.clone()
            q.grad = g.clone()
        if it == 1:                       # a scheduler changes the learning rate between steps
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 5e-3
        if it == 2:                       # a parameter without gradient is skipped by both
            pa[4].grad = None
            pb[4].grad = None
        norm = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        ob.step()
        oa.step()
        torch.testing.assert_close(oa.norm_clip[0], norm, rtol=1e-5, atol=0)
    for p, q in zip(pa, pb):
        torch.testing.assert_close(p, q, rtol=2e-5, atol=1e-6)      # fp32 rounding of one update (lr 1e-2, |p| ~ 1)
    for p, q in zip(pa, pb):
        torch.testing.assert_close(oa.state[p]["exp_avg"], ob.state[q]["exp_avg"], rtol=1e-5, atol=2e-7)
        torch.testing.assert_close(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"], rtol=1e-5, atol=1e-8)


def test_clip_adamw_state_dict_roundtrip():
    from mp_former_amd.optim import ClipAdamW
    dev = torch.device("cuda:0")
    ps = [t.clone().requires_grad_(True) for t in _make(1, dev)]
    o = ClipAdamW(ps, lr=1e-3, max_norm=0.5)
    for p in ps:
        p.grad = torch.ones_like(p)
    o.step()
    sd = o.state_dict()
    o2 = ClipAdamW(ps, lr=1e-3, max_norm=0.5)
    o2.load_state_dict(sd)
    assert float(o2.state[ps[0]]["step"]) == 1.0
    assert torch.equal(o2.state[ps[0]]["exp_avg"], o.state[ps[0]]["exp_avg"])


def test_clip_adamw_rejects_cpu():
    from mp_former_amd.optim import ClipAdamW
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        ClipAdamW([p]).step()
