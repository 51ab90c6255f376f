#!/usr/bin/env python3
"""Who issues the big layout copies / strided adds: torch.profiler with stacks, filtered by op name and input shape."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
opt = bench.build_optimizer(model)
batches = [bench.synth_batch(2, 1024, 80, i, dev) for i in range(2)]


def step(i):
    images, targets = batches[i % 2]
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(images, targets)
    loss.backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
names = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else {"aten::copy_", "aten::add", "aten::contiguous", "aten::clone"}
minel = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in names and ev.input_shapes and ev.input_shapes[0]:
        n = 1
        for d in ev.input_shapes[0]:
            n *= d
        if n >= minel:
            st = [s for s in ev.stack if "mp_former_amd" in s or "bench.py" in s][:3]
            print(f"{ev.name:18s} {str(ev.input_shapes[:2]):60s} dev {ev.device_time_total:8.1f} us  bwd={ev.is_async or ev.sequence_nr}  {st}")
