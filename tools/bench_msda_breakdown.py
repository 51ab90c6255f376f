#!/usr/bin/env python3
"""Per-kernel breakdown of the MSDA forward + backward at config B, N=2 (in-library HIP-event profiler).
usage: bench_msda_breakdown.py [init|trained|uniform] [cfg] [N] [-] [raw]     (4th argument: unused, kept for old command lines)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_msda import problem
from mp_former_amd import _lib, ms_deform_attn_backward, ms_deform_attn_forward, msda
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "init"
cfg = sys.argv[2] if len(sys.argv) > 2 else "B"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
value, shapes, lsi, loc, attn, go, S = problem(cfg, N, dev, mode)
ss = msda.attach_host_shapes(shapes, shapes.tolist(), lsi)
raw = len(sys.argv) > 5 and sys.argv[5] == "raw"          # the module-level (raw) form the encoder runs: grad_raw + amax slots
if raw:
    from mp_former_amd.gemm3 import amax_slots
    out = ms_deform_attn_forward(value, ss, lsi, loc, attn, 128)


def once():
    ms_deform_attn_forward(value, ss, lsi, loc, attn, 128)
    if raw:
        sl = amax_slots(2, dev) if os.environ.get("NO_AMAX") != "1" else (None, None)
        msda.ms_deform_attn_backward_raw(value, ss._mpf_host, loc, attn, go, out, sl[0], sl[1])
    else:
        ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)


for _ in range(3):
    once()
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(10):
    once()
torch.cuda.synchronize()
for k in ("msda_fwd", "msda_bwd_push", "msda_bwd_fill", "msda_bwd_pull", "msda_bwd_bin", "msda_sort_runs", "msda_bwd_tile"):
    n, ms, by = _lib.profile_get(k)
    if n:
        print(f"{k:14s} launches={n} avg={ms / n * 1e3:8.1f} us")
