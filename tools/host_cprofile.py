#!/usr/bin/env python3
"""cProfile of the launch thread over a few training steps of bench.py (python-level functions by own time): where the
python glue around the native calls spends the step's host time.  usage: host_cprofile.py [size] [top]"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mp_former_amd import _lib, _miopen  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
_lib.lib()
_miopen.use_shipped_find_db(check_version=True)
SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 512
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 45
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
opt = bench.build_optimizer(model)
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
for i in range(4):
    step(i)
torch.cuda.synchronize()
N = 5
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for v in st.stats.values())
print(f"python-visible time per step: {tot / N * 1e3:.2f} ms (profiler overhead included)")
for (fn, line, name), (cc, nc, tt, ct, _) in rows[:TOP]:
    print(f"{tt / N * 1e3:7.3f} ms own {ct / N * 1e3:7.3f} ms cum  x{nc / N:7.1f}  {os.path.basename(fn)}:{line} {name}")
if os.environ.get("MPF_CALLERS"):
    import io
    for pat in os.environ["MPF_CALLERS"].split(","):
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_callers(pat)
        lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
        print(f"---- callers of {pat} (counts over {N} steps)")
        print("\n".join(lines[:40]))
