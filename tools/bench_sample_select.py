#!/usr/bin/env python3
"""Micro-benchmark of the importance-sampling kernel at config B: 500 pairs, 256x256 bf16 planes, 3 x 12544 candidates."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib  # noqa: E402
from mp_former_amd.point_sample import MapSet, sample_select_uncertain  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
planes = torch.randn(1, 500, 256, 256, device=dev).to(torch.bfloat16)
ms = MapSet([planes])
offs = torch.arange(500, device=dev, dtype=torch.int64) * 65536
coords = torch.rand(500, 37632, 2, device=dev)
for _ in range(3):
    sample_select_uncertain(ms, offs, coords, 9408, 12544)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(10):
    sample_select_uncertain(ms, offs, coords, 9408, 12544)
torch.cuda.synchronize()
n, t, _ = _lib.profile_get("sample_select")
print(os.environ.get("MPF_LIB_PATH", "default"), n, round(t / n * 1e3, 1), "us")
