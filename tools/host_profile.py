#!/usr/bin/env python3
"""Host-side cost of one training step: torch.profiler CPU times by operator / autograd node
(where the launch thread spends its time)."""
import os, sys, time
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
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(images, targets)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 0.01, foreach=True)
    opt.step()

for i in range(4):
    step(i)
torch.cuda.synchronize()
if "--syncs" in sys.argv:
    torch.cuda.set_sync_debug_mode("warn")
    step(4)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    sys.exit(0)
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step(4)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in ka) / 1e3
print(f"total self CPU {tot:.1f} ms over {sum(e.count for e in ka)} events")
for e in rows[:45]:
    print(f"{e.self_cpu_time_total/1e3:8.2f} ms self  {e.cpu_time_total/1e3:8.2f} ms total  x{e.count:5d}  {e.key[:90]}")
