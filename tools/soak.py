#!/usr/bin/env python3
"""Soak run of the training step of bench.py: N steps over 4 fixed synthetic batches, loss every `--every` steps.
Evidence that the step TRAINS (losses fall on the repeated batches, nothing goes non-finite) — the parity tests pin
single steps, this pins the composition step -> optimizer -> step.  Run on the GPU box:
    python tools/soak.py --steps 400 > gpurun_out/soak.txt
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from mp_former_amd import _lib, _miopen
    _lib.lib()
    _miopen.use_shipped_find_db(check_version=True)
    from mp_former_amd import dropin
    dropin.configure_training_process()             # as bench.py: backward on the launch thread
    torch.manual_seed(0)
    model = bench.TrainModel().to(dev).train()
    model.backbone.to(memory_format=torch.channels_last)
    opt = bench.build_optimizer(model)
    batches = [bench.synth_batch(a.batch, a.size, 80, i, dev) for i in range(4)]
    hist = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        images, targets = batches[i % 4]
        opt.zero_grad(set_to_none=True)
        loss = model(images, targets)
        loss.backward()
        opt.step()
        if i % a.every == 0 or i == a.steps - 1:
            v = float(loss.detach())
            hist.append((i, v))
            print(f"step {i:5d}  loss {v:10.4f}", flush=True)
            assert v == v and abs(v) < 1e9, "non-finite loss"
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    first = sum(v for _, v in hist[:2]) / 2
    last = sum(v for _, v in hist[-2:]) / 2
    print(json.dumps({"steps": a.steps, "first": first, "last": last, "ratio": last / first, "s_total": round(dt, 1),
                      "params_finite": bool(all(torch.isfinite(p).all() for p in model.parameters()))}))


if __name__ == "__main__":
    main()
