#!/usr/bin/env python3
"""Per-step launch-thread times of bench.py's training step over many steps, with the cyclic garbage collector's pauses timed
(gc.callbacks): is the spread between runs of bench.py (23 ms against 29-33 ms for 20 steps on the same box) the collector?
usage: host_jitter.py [steps] [freeze]   (freeze = 1: gc.freeze() after the warm-up, as mp_former_amd.runtime.freeze_gc does)"""
import gc
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mp_former_amd import _lib, _miopen  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FREEZE = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
_lib.lib()
_miopen.use_shipped_find_db(check_version=True)
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
opt = bench.build_optimizer(model)
batches = [bench.synth_batch(2, 1024, 80, i, dev) for i in range(4)]


def step(i):
    images, targets = batches[i % 4]
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(images, targets)
    loss.backward()
    opt.step()


for i in range(8):
    step(i)
torch.cuda.synchronize()
if FREEZE:
    gc.collect()
    gc.freeze()
pauses, t_gc = [], [0.0]


def cb(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        pauses.append((info["generation"], (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(cb)
host = []
t_all = time.perf_counter()
for i in range(STEPS):
    t0 = time.perf_counter()
    step(i)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t_all) * 1e3 / STEPS
gc.callbacks.remove(cb)
hs = sorted(host)
print(f"freeze={int(FREEZE)} steps={STEPS}: wall {t_all:.2f} ms/step; host median {hs[len(hs) // 2]:.1f}, p90 {hs[int(len(hs) * 0.9)]:.1f}, max {hs[-1]:.1f} ms")
for gen in (0, 1, 2):
    p = [x for g_, x in pauses if g_ == gen]
    if p:
        print(f"  gc generation {gen}: {len(p)} collections, {sum(p):.1f} ms total, longest {max(p):.1f} ms")
print("  slowest steps: " + ", ".join(f"#{i}: {h:.1f}" for h, i in sorted(((h, i) for i, h in enumerate(host)), reverse=True)[:6]))
