#!/usr/bin/env python3
"""torch.profiler: which aten ops / autograd nodes launch the most kernels in one training step."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
opt = bench.build_optimizer(model)
params = [p for p in model.parameters() if p.requires_grad]
batches = [bench.synth_batch(2, 1024, 80, i, dev) for i in range(2)]
def step(i):
    images, targets = batches[i % 2]
    opt.zero_grad(set_to_none=True)
    loss = model(images, targets)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 0.01, foreach=True)
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(3); torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.self_device_time_total)[:45]
print(f"{'name':60s} {'calls':>6s} {'self_gpu_ms':>11s}")
for e in rows:
    print(f"{e.key[:60]:60s} {e.count:6d} {e.self_device_time_total/1e3:11.3f}")
