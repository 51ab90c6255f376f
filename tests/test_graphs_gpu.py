"""HIP-graph trunk (mp_former_amd/graphs.py): backbone + pixel decoder replayed as graphs give the eager step's loss and
gradients; small host tables travel as kernel arguments (mpf_upload_small)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,n", [(np.int64, 1), (np.int64, 496), (np.int32, 7), (np.float32, 992), (np.int64, 992), (np.int64, 993),
                                     (np.int32, 5000), (np.uint8, 12), (np.uint8, 13), (np.float64, 0)])
def test_upload_matches_source(dtype, n):
    """every size class of _h2d.upload: one and two kernel-argument launches, the pinned path above that, sizes that are not a
    multiple of four bytes, empty"""
    from mp_former_amd._h2d import upload
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + 1)
    a = (rng.integers(0, 250, size=n).astype(dtype) if np.issubdtype(dtype, np.integer) else rng.standard_normal(n).astype(dtype))
    t = upload(a, dev)
    assert t.device.type == "cuda" and tuple(t.shape) == (n,) and t.dtype == torch.from_numpy(a).dtype
    assert np.array_equal(t.cpu().numpy(), a)
    t2 = upload(a.reshape(-1, 1) if n else a, dev, torch.float32)
    assert t2.dtype == torch.float32 and np.array_equal(t2.cpu().numpy().reshape(-1), a.astype(np.float32))


def test_upload_inside_a_captured_graph_carries_the_table():
    from mp_former_amd._h2d import upload
    dev = torch.device("cuda:0")
    a = np.arange(300, dtype=np.int64)
    out = torch.zeros(300, dtype=torch.int64, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out.copy_(upload(a, dev) * 2)
    a[:] = -1                                   # the host array is gone / reused: the node keeps its own copy
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.arange(300) * 2)


def test_graphed_trunk_step_equals_eager_step():
    """bench.TrainModel at 256 x 256: the same batch through the eager path and through the graphed trunk (two steps each, so
    that a replay is compared and not only the capture's warm-up): equal loss, equal gradients up to the MSDA backward's
    entry order (fp32 reassociation inside a 4 x 4 tile)."""
    import bench
    from mp_former_amd import _miopen
    from mp_former_amd.graphs import GraphedTrunk
    dev = torch.device("cuda:0")
    _miopen.use_shipped_find_db(check_version=False)
    torch.manual_seed(3)
    model = bench.TrainModel().to(dev).train()
    model.backbone.to(memory_format=torch.channels_last)
    batches = [bench.synth_batch(2, 256, 80, 100 + i, dev) for i in range(2)]

    def run(trunk):
        model.trunk = trunk
        res = []
        for images, targets in batches:
            for p in model.parameters():
                p.grad = None
            torch.manual_seed(11)                # the point draws of matcher / criterion
            loss = model(images, targets)
            loss.backward()
            res.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        return res

    eager = run(None)
    trunk = GraphedTrunk(model.backbone, model.head.pixel_decoder, batches[0][0])
    graphed = run(trunk)
    model.trunk = None
    # (not bit-equal: MIOpen picks its bf16 convolution kernels per call context, and the backward has atomics-ordered sums)
    for (l0, g0), (l1, g1) in zip(eager, graphed):
        assert abs(l0 - l1) <= 1e-3 * abs(l0), (l0, l1)
        assert set(g0) == set(g1)
        errs = {k: float((g0[k] - g1[k]).norm() / (g0[k].norm() + 1e-20)) for k in g0}
        assert float(np.mean(list(errs.values()))) <= 2e-2 and max(errs.values()) <= 0.2, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
