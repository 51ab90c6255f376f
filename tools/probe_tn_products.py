#!/usr/bin/env python3
"""Probe: how much of a TN gemm3 launch is the six-product MFMA work?  The bf16-A variant of the same kernel (three products,
no split arithmetic, half the A bytes) against the fp32-A one on the encoder shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd.gemm3 import gemm3, gemm3_ex, split_weight  # noqa: E402
from tools.bench_gemm3 import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 43008
for (n, k) in ((256, 256), (1024, 256), (256, 1024)):
    a = torch.randn(M, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    planes = split_weight(w)
    a16 = a.bfloat16()
    t6 = timeit(lambda: gemm3(a, planes, b))
    t3 = timeit(lambda: gemm3_ex(a16, planes, b))
    print(f"N={n} K={k}: fp32 A (6 products) {t6:6.1f} us   bf16 A (3 products) {t3:6.1f} us", flush=True)
