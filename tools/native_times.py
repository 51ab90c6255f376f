#!/usr/bin/env python3
"""Per-kernel GPU time of the library's own kernels over full training steps (the in-library HIP-event
launch profiler; kernel time only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mp_former_amd import _lib

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

for i in range(3):
    step(i)
torch.cuda.synchronize()
_lib.profile_enable(True)
steps = 4
for i in range(steps):
    step(3 + i)
torch.cuda.synchronize()
names = ["msda_fwd", "msda_bwd_push", "tile_scan", "msda_bwd_fill", "msda_bwd_pull", "gemm3_tn_kernel<128>", "gemm3_tn_kernel<96>",
         "gemm3_nt_kernel<128>", "gemm3_nt_kernel<96>", "point_sample_kernel", "select_uncertain", "sample_select", "match_cost", "mask_loss_fwd", "point_sample_bits",
         "mask_loss_bwd", "attn_fwd", "attn_bwd_kv", "attn_bwd_q", "attn_mask", "res_ln256_fwd", "res_ln256_bwd", "bias_act"]
tot = 0.0
for n in names:
    c, ms, by = _lib.profile_get(n)
    if c:
        tot += ms
        print(f"{n:24s} {c / steps:6.1f} launches/step  {ms / steps:7.3f} ms/step  avg {ms / c * 1e3:8.1f} us" + (f"  {by / ms / 1e6:7.1f} GB/s alg" if by else ""))
print(f"listed kernels: {tot / steps:.2f} ms/step")
