#!/usr/bin/env python3
"""Experiment: the backbone (static shapes at the fixed 1024 x 1024 crop) as HIP graphs — one replay for its forward, one for
its backward (torch.cuda.make_graphed_callables) — against the eager backbone: step time, host enqueue time, loss trajectory."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
from mp_former_amd import _lib, _miopen  # noqa: E402
_lib.lib()
_miopen.use_shipped_find_db(check_version=True)


class TupleBackbone(torch.nn.Module):
    def __init__(self, bb):
        super().__init__()
        self.bb = bb

    def forward(self, x):
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            f = self.bb(x)
        return tuple(f[k] for k in ("res2", "res3", "res4", "res5"))


def run(graphed):
    torch.manual_seed(0)
    model = bench.TrainModel().to(dev).train()
    model.backbone.to(memory_format=torch.channels_last)
    opt = bench.build_optimizer(model)
    batches = [bench.synth_batch(2, 1024, 80, i, dev) for i in range(4)]
    tb = TupleBackbone(model.backbone)
    call = tb
    if graphed:
        import functools
        _orig = torch.cuda.graph.__init__
        if not getattr(torch.cuda.graph, "_mpf_relaxed", False):
            torch.cuda.graph.__init__ = functools.partialmethod(_orig, capture_error_mode=os.environ.get("MPF_CAPTURE_MODE", "relaxed"))
            torch.cuda.graph._mpf_relaxed = True
        sample = batches[0][0].contiguous(memory_format=torch.channels_last).clone()
        call = torch.cuda.make_graphed_callables(tb, (sample,), num_warmup_iters=3, allow_unused_input=True)

    def step(i):
        images, targets = batches[i % 4]
        opt.zero_grad(set_to_none=True)
        feats = call(images.contiguous(memory_format=torch.channels_last))
        feats = dict(zip(("res2", "res3", "res4", "res5"), feats))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model.head.total_loss(feats, targets)
        loss.backward()
        opt.step()
        return loss

    losses = []
    for i in range(6):
        losses.append(float(step(i)))
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(30):
        t1 = time.perf_counter()
        step(i)
        host.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 30
    print(f"graphed={graphed}: {ms:.2f} ms/step, host enqueue median {sorted(host)[15]:.1f} ms, first losses {[round(x, 3) for x in losses[:4]]}", flush=True)


for g in (False, True, False, True):
    run(g)
