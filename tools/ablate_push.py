#!/usr/bin/env python3
"""Ablation timings of the MSDA push kernel (benchmark only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_msda import problem
from mp_former_amd import _lib, ms_deform_attn_backward, msda
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "init"
value, shapes, lsi, loc, attn, go, S = problem("B", 2, dev, mode)
ss = msda.attach_host_shapes(shapes, shapes.tolist(), lsi)
for ab in (0, 7, 15, 23, 31, 8, 16):
    _lib.set_option("msda_push_ablate2", ab)
    for _ in range(3):
        ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10):
        ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
    torch.cuda.synchronize()
    n, ms, _ = _lib.profile_get("msda_bwd_push")
    n2, ms2, _ = _lib.profile_get("msda_bwd_pull")
    print(f"ablate={ab}: push {ms / n * 1e3:7.1f} us   pull {ms2 / n2 * 1e3:7.1f} us")
    _lib.profile_enable(False)
_lib.set_option("msda_push_ablate2", 0)
