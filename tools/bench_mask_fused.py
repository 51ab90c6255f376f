#!/usr/bin/env python3
"""Micro-benchmark of the fused mask kernels at config B (10 outputs, N = 2, 100 + 14 queries, 256 x 256 features, 12 544 points,
~500 pairs): matching cost from the factors, pair planes forward / backward."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib, mask_fused  # noqa: E402
from mp_former_amd._h2d import upload  # noqa: E402
from mp_former_amd.criterion import SetCriterion  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
L, N, Q, Qt, H, W, P = 10, 2, 100, 114, 256, 256, 12544
counts = np.array([5, 14])
me = (torch.randn(L * Qt, N, 256) * 0.15).to(torch.bfloat16).to(dev).transpose(0, 1)
mf = torch.randn(N, H, W, 256).to(torch.bfloat16).to(dev).permute(0, 3, 1, 2)
root = mask_fused.FactoredMasks(me, mf)
views = [root[:, l * Qt:(l + 1) * Qt][:, -Q:] for l in range(L)]
Tt, Tmax = int(counts.sum()), int(counts.max())
coords = torch.rand(L * N, P, 2, device=dev)
tsamp = (torch.rand(L * Tt, P, device=dev) > 0.5).float()
firsts = np.concatenate([[0], np.cumsum(counts)[:-1]])
l_idx, b_idx = [x.reshape(-1) for x in np.meshgrid(np.arange(L), np.arange(N), indexing="ij")]


def cost():
    return mask_fused.match_cost_fused(views, coords, tsamp, l_idx * Tt + firsts[b_idx], counts[b_idx], l_idx, b_idx, Q, Tmax, 5.0, 5.0)


# ~500 pairs: per output, per image T matched + T mask-piloted rows
bi = np.concatenate([np.full(2 * c, b) for _ in range(L) for b, c in enumerate(counts)])
rows = np.concatenate([l * Qt + np.arange(2 * c) for l in range(L) for b, c in enumerate(counts)])
n = len(bi)
slot, first, count = SetCriterion._slot_layout(bi, N)
inv = np.zeros(n, dtype=np.int32)
inv[slot] = np.arange(n, dtype=np.int32)
row_off = upload(bi * me.stride(0) + rows * me.stride(1), dev)
i32 = upload(np.concatenate([inv, first, count]).astype(np.int32), dev)
a = me.detach().clone().requires_grad_(True)
b = mf.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
g = torch.randn(n, H * W, device=dev).to(torch.bfloat16)


def planes():
    p = mask_fused.PairPlanes.apply(a, b, row_off, i32[:n], i32[n:n + N], i32[n + N:], n, int(count.max()))
    p.backward(g)
    a.grad = None
    b.grad = None


for _ in range(3):
    cost(); planes()
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(10):
    cost(); planes()
torch.cuda.synchronize()
out = [os.environ.get("MPF_LIB_PATH", "default").split("/")[-1], f"pairs={n}"]
for k in ("match_cost_fused", "pair_planes_fwd", "pair_planes_dfeat", "pair_planes_dembed"):
    c, t, _ = _lib.profile_get(k)
    out.append(f"{k} {t / max(c, 1) * 1e3:.1f} us")
print("  ".join(out))
