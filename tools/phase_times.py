#!/usr/bin/env python3
"""Per-phase GPU time of the training step (events on the current stream) — optimisation guide."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = bench.TrainModel().to(dev).train()
    model.backbone.to(memory_format=torch.channels_last)
    opt = bench.build_optimizer(model)
    params = [p for p in model.parameters() if p.requires_grad]
    batches = [bench.synth_batch(2, 1024, 80, i, dev) for i in range(2)]
    names = ["backbone_fwd", "pixdec_fwd", "decoder_fwd", "criterion_fwd", "backward", "clip", "adamw"]
    bw_names = ["bwd: criterion+decoder", "bwd: pixel decoder", "bwd: backbone"]
    bw_acc = {n: 0.0 for n in bw_names}
    acc = {n: 0.0 for n in names}
    cpu = {n: 0.0 for n in names}
    wall = 0.0
    iters = 6
    for it in range(iters):
        images, targets = batches[it % 2]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        ct = [0.0] * (len(names) + 1)
        ev[0].record(); ct[0] = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats = model.backbone(images.contiguous(memory_format=torch.channels_last))
            ev[1].record(); ct[1] = time.perf_counter()
            mf, _, ms = model.head.pixel_decoder.forward_features(feats)
            ev[2].record(); ct[2] = time.perf_counter()
            bev = {"dec": torch.cuda.Event(enable_timing=True), "pix": torch.cuda.Event(enable_timing=True)}
            cnt = {"dec": 0, "pix": 0}

            def mk(tag, total):
                def hook(g):
                    cnt[tag] += 1
                    if cnt[tag] == total:
                        bev[tag].record()
                    return g
                return hook
            for t_ in [mf] + list(ms):
                t_.register_hook(mk("dec", 1 + len(ms)))
            fl = [v for v in feats.values() if v.requires_grad]
            for t_ in fl:
                t_.register_hook(mk("pix", len(fl)))
            out = model.head.predictor(ms, mf, None, {"tgt": targets, "scalar": 1, "noise_scale": 0.0})
            ev[3].record(); ct[3] = time.perf_counter()
            losses = model.head.criterion(out, targets)
            loss = model.head.criterion.weighted_total(losses)
        ev[4].record(); ct[4] = time.perf_counter()
        loss.backward()
        ev[5].record(); ct[5] = time.perf_counter()
        torch.nn.utils.clip_grad_norm_(params, 0.01, foreach=True)
        ev[6].record(); ct[6] = time.perf_counter()
        opt.step()
        ev[7].record(); ct[7] = time.perf_counter()
        torch.cuda.synchronize()
        if it >= 2:
            wall += time.perf_counter() - t0
            for i, n in enumerate(names):
                acc[n] += ev[i].elapsed_time(ev[i + 1])
                cpu[n] += (ct[i + 1] - ct[i]) * 1e3
            bw_acc[bw_names[0]] += ev[4].elapsed_time(bev["dec"])
            bw_acc[bw_names[1]] += bev["dec"].elapsed_time(bev["pix"])
            bw_acc[bw_names[2]] += bev["pix"].elapsed_time(ev[5])
    n = iters - 2
    print("phase times (ms / step, 2 images):    GPU-timeline   host (launch) time")
    for k in names:
        print(f"  {k:14s} {acc[k] / n:8.2f}   {cpu[k] / n:8.2f}")
    for k in bw_names:
        print(f"    {k:24s} {bw_acc[k] / n:8.2f}")
    print(f"  {'sum':14s} {sum(acc.values()) / n:8.2f}   wall {wall / n * 1e3:8.2f}")


if __name__ == "__main__":
    main()
