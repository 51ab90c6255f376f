#!/usr/bin/env python3
"""Kernel-level profile of the decoder forward (+ its backward) alone, config B, N=2."""
import os, sys, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.TrainModel().to(dev).train()
images, targets = bench.synth_batch(2, 1024, 80, 0, dev)
with torch.autocast("cuda", dtype=torch.bfloat16):
    feats = model.backbone(images)
    mf, _, ms = model.head.pixel_decoder.forward_features(feats)
mf = mf.detach().requires_grad_(True); ms = [m.detach().requires_grad_(True) for m in ms]
dn = {"tgt": targets, "scalar": 1, "noise_scale": 0.0}
def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return model.head.predictor(ms, mf, None, dn)
def run(which):
    out = fwd()
    if which == "bwd":
        loss = sum(o["pred_masks"].float().mean() + o["pred_logits"].float().mean() for o in [out] + out["aux_outputs"])
        loss = loss + sum(o["pred_masks"].float().mean() + o["pred_logits"].float().mean() for o in [out["dn_out"]] + out["dn_out"]["aux_outputs"])
        loss.backward()
for _ in range(3): run("bwd")
torch.cuda.synchronize()
for which in ("fwd", "bwd"):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run(which); torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.self_device_time_total > 0]
    tot = sum(e.self_device_time_total for e in ev) / 1e3
    n = sum(e.count for e in ev)
    print(f"== decoder {which}: {tot:.2f} ms GPU busy, {n} kernels")
    for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:22]:
        print(f"  {e.self_device_time_total/1e3:7.3f} ms x{e.count:4d} avg {e.self_device_time_total/e.count:7.1f} us  {e.key[:100]}")
