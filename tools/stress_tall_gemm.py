#!/usr/bin/env python3
"""Repeatability stress of mpf_tall_gemm_bf16: the same product many times, concurrently with a memory-bound kernel on a second
stream (so that load / store completion order varies), every result compared bit for bit with the first."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from mp_former_amd.small_linear import tall_gemm  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
side = torch.cuda.Stream()
junk = torch.randn(64 << 20, device=dev)
for (M, K, N) in ((32768, 256, 768), (8192, 256, 256), (2400, 256, 256), (32768, 768, 256), (8192, 768, 256)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    ref = tall_gemm(x, w, b).clone()
    n_bad = 0
    for i in range(300):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)
        y = tall_gemm(x, w, b)
        if not torch.equal(y, ref):
            n_bad += 1
    torch.cuda.synchronize()
    print(f"M {M} K {K} N {N}: {n_bad} of 300 runs differ from the first", flush=True)
    bad += n_bad
print("TOTAL differing runs:", bad)
