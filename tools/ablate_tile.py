#!/usr/bin/env python3
"""Where the MSDA tile kernel's time goes (config B, N = 2): mpf_set_option("msda_push_ablate2", bits) — 1 no corner dot
products, 2 no MFMA role, 4 no grad_out row DMA, 8 synthetic locations instead of the loc / attn gathers.
usage: ablate_tile.py [init|trained]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_msda import problem
from mp_former_amd import _lib, ms_deform_attn_backward, msda
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "init"
value, shapes, lsi, loc, attn, go, S = problem("B", 2, dev, mode)
ss = msda.attach_host_shapes(shapes, shapes.tolist(), lsi)
for bits in (0, 1, 2, 4, 8, 3, 5, 6, 7, 15, 0):
    _lib.set_option("msda_push_ablate2", bits)
    for _ in range(3):
        ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10):
        ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
    torch.cuda.synchronize()
    n, ms, _ = _lib.profile_get("msda_bwd_tile")
    nb, msb, _ = _lib.profile_get("msda_bwd_bin")
    _lib.profile_enable(False)
    print(f"ablate {bits:2d}: tile {ms / n * 1e3:7.1f} us   bin {msb / nb * 1e3:6.1f} us")
_lib.set_option("msda_push_ablate2", 0)
