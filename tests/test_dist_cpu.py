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


def test_world1_is_identity():
    from mp_former_amd import dist as mdist
    assert mdist.world_size() == 1
    assert mdist.global_num_masks(0, torch.device("cpu")) == 1.0
    assert mdist.global_num_masks(7, torch.device("cpu")) == 7.0
    m = torch.nn.Linear(2, 2)
    assert mdist.wrap_ddp(m) is m
