#!/usr/bin/env python3
"""match_cost kernel time vs number of targets (config B sizes: 2000 rows of 256x256 bf16 planes, P = 12544)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib
from mp_former_amd.point_sample import MapSet, match_cost

dev = torch.device("cuda:0")
L, N, Q, P = 10, 2, 100, 12544
maps = [torch.randn(N, Q, 256, 256, device=dev).to(torch.bfloat16) for _ in range(L)]
ms = MapSet(maps)
l_idx, b_idx, q_idx = [a.reshape(-1) for a in np.meshgrid(np.arange(L), np.arange(N), np.arange(Q), indexing="ij")]
offs = torch.from_numpy(ms.offsets(l_idx, b_idx, q_idx)).to(dev)
coords = torch.rand(L * N, P, 2, device=dev)
crow = torch.from_numpy((l_idx * N + b_idx).astype(np.int32)).to(dev)
for T in (1, 4, 10, 20):
    tsamp = torch.rand(L * N * T, P, device=dev)
    tfirst = torch.from_numpy(((l_idx * N + b_idx) * T).astype(np.int32)).to(dev)
    tcount = torch.full((L * N * Q,), T, dtype=torch.int32, device=dev)
    for grp in (1, Q):
        f = lambda: match_cost(ms, offs, coords, crow, tsamp, tfirst, tcount, T, 5.0, 5.0, rows_per_group=grp)
        for _ in range(2):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"T={T:2d} rows_per_group={grp:3d} ({_lib.last_kernel()}): {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)
