#!/usr/bin/env python3
"""Which element-wise ops (with shapes) the training step still runs: torch.profiler with shapes, GPU time per (op, shapes)."""
import os, sys, collections
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(4)
    torch.cuda.synchronize()
want = ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zero_", "aten::mul", "aten::threshold_backward", "aten::sum",
        "aten::clamp_min", "aten::relu", "aten::cat", "aten::div", "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::mul_")
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if "--aten" in sys.argv:
        if e.key.startswith("aten::") and e.self_device_time_total > 0 and "convolution" not in e.key:
            rows.append((e.self_device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:150]))
    elif (e.key in want or "--all" in sys.argv) and e.self_device_time_total > 0:
        rows.append((e.self_device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for t, c, k, sh in rows[:(200 if '--aten' in sys.argv else 70 if '--all' in sys.argv else 45)]:
    print(f"{t:7.3f} ms x{c:4d}  {k:26s} {sh}")
