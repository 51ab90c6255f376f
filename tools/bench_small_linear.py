#!/usr/bin/env python3
"""Small-row bf16 Linear: kernel time of forward / dX / dW(+db) vs the library on the decoder shapes."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd.small_linear import small_gemm  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    for (M, K, N, relu) in ((228, 256, 256, False), (228, 256, 2048, True), (228, 2048, 256, False), (228, 256, 768, False)):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        g = torch.randn(M, N, device=dev).bfloat16()
        y = F.linear(x, w, b).relu() if relu else None
        t_f = timeit(lambda: small_gemm(x, K, 1, w, K, 1, M, N, K, bias=b, relu=relu))
        t_dx = timeit(lambda: small_gemm(g, N, 1, w, 1, K, M, K, N, gate=y))
        t_dw = timeit(lambda: small_gemm(g, 1, N, x, 1, K, N, K, M, gate=y, rowsum=True))
        l_f = timeit(lambda: F.linear(x, w, b))
        l_dx = timeit(lambda: g @ w)
        l_dw = timeit(lambda: (g.t() @ x, g.sum(0)))
        print(f"M={M} K={K} N={N} relu={relu}: fwd {t_f:5.1f} (lib {l_f:5.1f})  dX {t_dx:5.1f} (lib {l_dx:5.1f})  "
              f"dW+db {t_dw:5.1f} (lib {l_dw:5.1f}) us  [back-to-back launches, includes host issue rate]", flush=True)


if __name__ == "__main__":
    main()
