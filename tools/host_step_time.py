#!/usr/bin/env python3
"""Host time to ENQUEUE one training step of bench.py (no synchronisation inside the timed region) next to the GPU time of
the step: the launch thread must stay ahead of the device for the step time to be the GPU's."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
from mp_former_amd import _lib, _miopen  # noqa: E402
_lib.lib()
_miopen.use_shipped_find_db(check_version=True)
if os.environ.get("MPF_CONV_FIND", "0") == "1":          # (experiment: the non-immediate MIOpen path with its per-process algorithm cache)
    torch.backends.cudnn.benchmark = True
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
opt = bench.build_optimizer(model)
SIZE = int(os.environ.get("MPF_SIZE", "1024"))
batches = [bench.synth_batch(2, SIZE, 80, i, dev) for i in range(2)]


def step(i):
    images, targets = batches[i % 2]
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(images, targets)
    loss.backward()
    opt.step()


from mp_former_amd import dropin
dropin.configure_training_process(single_thread_autograd=True)      # backward on the calling thread, as bench.py
for i in range(5):
    step(i)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for i in range(10):
    t0 = time.perf_counter()
    step(i)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t_all) * 1e3 / 10
print(f"host enqueue per step: {sorted(host)[len(host) // 2]:.1f} ms (min {min(host):.1f}, max {max(host):.1f}); wall per step {t_all:.1f} ms")
