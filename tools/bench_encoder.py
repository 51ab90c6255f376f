#!/usr/bin/env python3
"""Encoder-only timing (6 MSDeformAttn layers, config B, N=2): forward and forward+backward GPU time,
with the per-kernel launch table of one fwd+bwd.  Usage: bench_encoder.py [--fused 0|1] [--table]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import pixel_decoder as PD  # noqa: E402


def run(fused, iters=10, table=False, size=1024, batch=2):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if fused is not None:
        os.environ["MPF_FUSED_ENCODER"] = "1" if fused else "0"
    enc = PD.MSDeformAttnTransformerEncoderOnly(d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024,
                                                dropout=0.0, num_feature_levels=3).to(dev).train()
    with torch.no_grad():
        for p in enc.parameters():      # non-degenerate offsets / attention logits
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                p.normal_(0, 0.02)
    srcs = [torch.randn(batch, 256, size // s, size // s, device=dev, requires_grad=True) for s in (32, 16, 8)]
    pe = PD.PositionEmbeddingSine(128, normalize=True)
    pos = [pe(s) for s in srcs]

    def step(bwd):
        mem, _, _ = enc(srcs, pos)
        if bwd:
            mem.backward(go)
        return mem

    go = torch.randn(batch, sum((size // s) ** 2 for s in (32, 16, 8)), 256, device=dev)
    for _ in range(3):
        step(True)
    out = {}
    for name, bwd in (("fwd", False), ("fwd+bwd", True)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            step(bwd)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters
    print(f"encoder fused={fused}: fwd {out['fwd']:.3f} ms  fwd+bwd {out['fwd+bwd']:.3f} ms", flush=True)
    if table:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            step(True)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", type=int, default=None)
    ap.add_argument("--table", action="store_true")
    ap.add_argument("--both", action="store_true")
    a = ap.parse_args()
    if a.both:
        run(0, table=a.table)
        run(1, table=a.table)
    else:
        run(a.fused, table=a.table)
