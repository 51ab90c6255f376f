"""CPU, world_size 2, gloo: the N>1 plumbing of the hot path (mp_former_amd/dist.py) — the
`num_masks` all-reduce of criterion.py:235-237, max-over-ranks timing, and DDP gradient averaging as
bench.py / Detectron2 set it up.  The path shards by image only, so there is no other collective."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mp_former_amd import dist as mdist
    r, w = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    out = {}
    # num_masks: rank 0 has 3 GT masks, rank 1 has 8 -> (3+8)/2 = 5.5 on both ranks
    out["num_masks"] = mdist.global_num_masks([3, 8][rank], dev)
    # clamp: no GT anywhere -> 1
    out["num_masks_empty"] = mdist.global_num_masks(0, dev)
    out["num_masks_dev"] = float(mdist.global_num_masks_device([3, 8][rank], dev))      # the no-sync form of the criterion
    out["max"] = mdist.max_over_ranks([0.25, 0.75][rank], dev)
    # DDP: gradients are averaged over ranks
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    ddp = mdist.wrap_ddp(model)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    x = torch.full((2, 8), float(rank + 1))
    ddp(x).sum().backward()
    out["grad"] = model[0].weight.grad.tolist()
    # reference: average of the two single-rank gradients
    ref = torch.zeros_like(model[0].weight.grad)
    for rr in range(world):
        m2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        m2.load_state_dict(model.state_dict())
        m2(torch.full((2, 8), float(rr + 1))).sum().backward()
        ref += m2[0].weight.grad / world
    out["grad_ref"] = ref.tolist()
    # FlatGradSync as bench.py drives it: THREE flat buckets — head | late backbone stage | early backbone stages — the first
    # two launched from tensor hooks on the feature maps while the rest of the backbone back-propagates, the last at
    # finish(); parameters that differ per rank before the broadcast, an unused parameter, channels-last parameters
    torch.manual_seed(10 + rank)                       # different initial weights per rank: the broadcast must fix that

    def make():
        bb = torch.nn.ModuleDict({"stem": torch.nn.Conv2d(3, 4, 3, padding=1), "early": torch.nn.Conv2d(4, 4, 3, padding=1),
                                  "late": torch.nn.Conv2d(4, 4, 3, padding=1)})
        hd = torch.nn.ModuleDict({"a": torch.nn.Linear(4, 5), "b": torch.nn.Linear(4, 5), "unused": torch.nn.Linear(2, 2)})
        return bb, hd

    backbone, head = make()
    backbone = backbone.to(memory_format=torch.channels_last)
    names = lambda bb, hd: list(hd.named_parameters()) + list(bb.named_parameters())  # noqa: E731
    sync = mdist.FlatGradSync([list(head.parameters()), list(backbone["late"].parameters()),
                               list(backbone["stem"].parameters()) + list(backbone["early"].parameters())])
    out["w0_after_broadcast"] = backbone["stem"].weight.detach().flatten()[:4].tolist()
    launched = []

    def net(bb, hd, xin, hooks):
        f_early = torch.relu(bb["early"](torch.relu(bb["stem"](xin))))
        f_late = torch.relu(bb["late"](f_early))
        if hooks:   # the head's nodes are created last, so autograd runs them before the first backbone node
            f_late.register_hook(lambda g: (launched.append(0), sync.launch(0))[2:] or None)
            f_early.register_hook(lambda g: (launched.append(1), sync.launch(1))[2:] or None)
        return (hd["a"](f_late.mean((2, 3))) + hd["b"](f_early.mean((2, 3)))).square().sum()

    xin = torch.full((2, 3, 6, 6), float(rank + 1)) + torch.arange(6.0).view(1, 1, 1, 6)
    net(backbone, head, xin, True).backward()
    assert sync.groups[0]["launched"] and sync.groups[1]["launched"] and not sync.groups[2]["launched"]
    sync.finish()
    out["flat_launched_early"] = list(launched)
    out["flat_grads"] = {n: p.grad.flatten().tolist() for n, p in names(backbone, head)}
    out["flat_strides_ok"] = all(p.grad.stride() == p.stride() for _, p in names(backbone, head))
    refg = {}
    for rr in range(world):
        b2, h2 = make()
        b2.load_state_dict(backbone.state_dict()); h2.load_state_dict(head.state_dict())
        xr = torch.full((2, 3, 6, 6), float(rr + 1)) + torch.arange(6.0).view(1, 1, 1, 6)
        net(b2, h2, xr, False).backward()
        for n, p in names(b2, h2):
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            refg[n] = refg.get(n, 0) + g.flatten() / world
    out["flat_ref"] = {n: v.tolist() for n, v in refg.items()}
    # a second step reuses the buckets (p.grad is the view now; zero_grad(set_to_none) as the bench does)
    for _, p in names(backbone, head):
        p.grad = None
    net(backbone, head, xin, True).backward()
    sync.finish()
    out["flat_grads_step2"] = {n: p.grad.flatten().tolist() for n, p in names(backbone, head)}
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert res[r]["num_masks"] == pytest.approx(5.5)
        assert res[r]["num_masks_empty"] == 1.0
        assert res[r]["num_masks_dev"] == pytest.approx(5.5)
        assert res[r]["max"] == pytest.approx(0.75)
        torch.testing.assert_close(torch.tensor(res[r]["grad"]), torch.tensor(res[r]["grad_ref"]))
    torch.testing.assert_close(torch.tensor(res[0]["grad"]), torch.tensor(res[1]["grad"]))
    assert res[0]["w0_after_broadcast"] == res[1]["w0_after_broadcast"]
    for r in range(2):
        assert res[r]["flat_launched_early"] == [0, 1] and res[r]["flat_strides_ok"]     # the head bucket first, then the late stage
        assert set(res[r]["flat_grads"]) == set(res[r]["flat_ref"])
        for n in res[r]["flat_ref"]:
            torch.testing.assert_close(torch.tensor(res[r]["flat_grads"][n]), torch.tensor(res[r]["flat_ref"][n]), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(torch.tensor(res[r]["flat_grads_step2"][n]), torch.tensor(res[r]["flat_ref"][n]), rtol=1e-5, atol=1e-6)
        assert all(v == 0.0 for v in res[r]["flat_grads"]["unused.weight"])
    assert res[0]["flat_grads"] == res[1]["flat_grads"]


def _wire_worker(rank, world, port, q):
    """5 clipped AdamW steps of a small two-bucket model, gradients averaged by FlatGradSync over an fp32 and over a bf16 wire
    (the reference's hook point: train_net.py:307-322 clips the AVERAGED gradients, then AdamW)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mp_former_amd import dist as mdist
    mdist.init_from_env("gloo")
    out = {}
    for name, wire in (("fp32", None), ("bf16", torch.bfloat16)):
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.GELU(), torch.nn.Linear(32, 32), torch.nn.GELU(), torch.nn.Linear(32, 5))
        p0 = [p.detach().clone() for p in net.parameters()]
        sync = mdist.FlatGradSync([list(net[4].parameters()) + list(net[2].parameters()), list(net[0].parameters())], wire_dtype=wire)
        slots = [sum((p.numel() + 63) // 64 * 64 for p in g["params"]) for g in sync.groups]
        assert sync.bucket_bytes() == [n * (2 if wire else 4) for n in slots]          # half the bytes on the bf16 wire
        opt = torch.optim.AdamW(net.parameters(), lr=1e-2, weight_decay=0.05)
        gen = torch.Generator().manual_seed(100 + rank)                  # every rank its own images
        wire_exact = True
        for step in range(5):
            for p in net.parameters():
                p.grad = None
            x = torch.randn(16, 12, generator=gen)
            y = torch.randn(16, 5, generator=gen)
            (net(x) - y).square().mean().backward()
            sync.launch(0)                                               # (what the hook does under the backbone's backward)
            sync.finish()
            if wire is not None:                                         # the averaged gradient is a sum of bf16 numbers / world
                for g in sync.groups:
                    wire_exact &= bool(torch.equal(g["flat"], g["wire"].float() / world))
            torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
            opt.step()
        out[name] = [p.detach().clone() for p in net.parameters()]
        out[name + "_delta"] = [a - b for a, b in zip(out[name], p0)]
        out[name + "_wire_exact"] = wire_exact
    q.put((rank, {k: ([t.tolist() for t in v] if isinstance(v, list) else v) for k, v in out.items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_bf16_wire_drift_is_bounded():
    """SURVEY.md 8 f4 (second half): bf16 gradient all-reduce.  Over 5 clipped AdamW steps the parameters reached through the bf16
    wire stay identical on both ranks and within 0.5 % (relative L2 of the total update, per tensor) of those reached through the
    fp32 wire — bf16 rounds each rank's gradient to 8 bits (2^-9 relative), Adam's normalised update turns that into a
    same-order perturbation of the step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["bf16"] == res[1]["bf16"] and res[0]["fp32"] == res[1]["fp32"]          # replicas stay bit-identical
    assert res[0]["bf16_wire_exact"] and res[1]["bf16_wire_exact"]
    assert res[0]["bf16"] != res[0]["fp32"]                                                # (the wire really was bf16)
    worst = 0.0
    for d16, d32 in zip(res[0]["bf16_delta"], res[0]["fp32_delta"]):
        d16, d32 = torch.tensor(d16), torch.tensor(d32)
        worst = max(worst, float((d16 - d32).norm() / d32.norm()))
    assert worst < 5e-3, worst          # measured 1.7e-3
    print("bf16-wire drift, worst tensor, relative to the 5-step update:", worst)


def test_flat_grad_sync_wire_dtype_world1():
    """world size 1 (no process group): the bf16 wire still rounds the gradients once (launch casts, finish unpacks)"""
    from mp_former_amd import dist as mdist
    lin = torch.nn.Linear(5, 3)
    sync = mdist.FlatGradSync([list(lin.parameters())], broadcast=False, wire_dtype=torch.bfloat16)
    lin(torch.randn(4, 5)).square().sum().backward()
    g = lin.weight.grad.clone()
    sync.finish()
    assert torch.equal(lin.weight.grad, g.bfloat16().float())
    with pytest.raises(ValueError):
        mdist.FlatGradSync([list(lin.parameters())], broadcast=False, wire_dtype=torch.int8)


def test_flat_grad_sync_rejects_late_gradients():
    """a gradient that shows up for a bucket after its all-reduce was launched must not be dropped silently"""
    from mp_former_amd import dist as mdist
    lin = torch.nn.Linear(3, 2)
    sync = mdist.FlatGradSync([list(lin.parameters())], broadcast=False)
    lin(torch.ones(1, 3)).sum().backward()
    sync.launch(0)
    lin(torch.ones(1, 3)).sum().backward()          # arrives after the launch
    with pytest.raises(RuntimeError, match="arrived after"):
        sync.finish()


def test_world1_is_identity():
    from mp_former_amd import dist as mdist
    assert mdist.world_size() == 1
    assert mdist.global_num_masks(0, torch.device("cpu")) == 1.0
    assert mdist.global_num_masks(7, torch.device("cpu")) == 7.0
    m = torch.nn.Linear(2, 2)
    assert mdist.wrap_ddp(m) is m
