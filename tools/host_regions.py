#!/usr/bin/env python3
"""Host (launch-thread) time of one training step by region: each region is entered with the device idle and timed until the
call RETURNS (no synchronisation inside), at a small image size where the GPU is never the one waited for."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
from mp_former_amd import _lib, _miopen  # noqa: E402
_lib.lib()
_miopen.use_shipped_find_db(check_version=True)
SIZE = int(os.environ.get("MPF_SIZE", "256"))
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
opt = bench.build_optimizer(model)
batches = [bench.synth_batch(2, SIZE, 80, i, dev) for i in range(2)]
acc = {}


def region(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


def step(i, timed):
    images, targets = batches[i % 2]
    reg = region if timed else (lambda n, f: f())
    reg("zero_grad", lambda: opt.zero_grad(set_to_none=True))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = reg("backbone fwd", lambda: model.backbone(images.contiguous(memory_format=torch.channels_last)))
        h = model.head
        pd = reg("pixel decoder fwd", lambda: h.pixel_decoder.forward_features(feats))
        mask_features, _, multi_scale = pd
        dn_args = {"tgt": targets, "scalar": h.scalar, "noise_scale": h.noise_scale}
        outputs = reg("decoder fwd", lambda: h.predictor(multi_scale, mask_features, None, dn_args))
        loss = reg("criterion fwd", lambda: h.criterion.weighted_total(h.criterion(outputs, targets)))
    reg("backward (all)", lambda: loss.backward())
    reg("optimizer", lambda: opt.step())


for i in range(4):
    step(i, False)
n = 6
for i in range(n):
    step(i, True)
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:20s} {v / n:7.2f} ms")
print(f"{'total':20s} {tot / n:7.2f} ms  (size {SIZE})")
