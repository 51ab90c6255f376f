#!/usr/bin/env python3
"""Per-kernel breakdown of the binned MSDA backward (in-library HIP-event profiler)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_msda import problem
from mp_former_amd import _lib, ms_deform_attn_backward, msda
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "init"
value, shapes, lsi, loc, attn, go, S = problem("B", 2, dev, mode)
msda.BWD_MODE = "binned"
if len(sys.argv) > 2:
    _lib.set_option("msda_push_ablate", int(sys.argv[2]))
for _ in range(3):
    ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 128)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(10):
    ms_deform_attn_backward(value, shapes, lsi, loc, attn, go, 128)
torch.cuda.synchronize()
for k in ("push", "fill", "pull"):
    n, ms, by = _lib.profile_get("msda_bwd_" + k)
    print(f"{k:5s} launches={n} avg={ms / max(n, 1) * 1e3:8.1f} us")
