#!/usr/bin/env python3
"""mpf_tall_gemm_bf16 at the key / value projection shapes of the decoder (config B, N = 2) vs the library's F.linear."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mp_former_amd.small_linear import tall_gemm  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot = [0.0, 0.0]
for M in (32768, 8192, 2048):
    for (K, N) in ((256, 768), (768, 256), (256, 96)):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        t0, t1 = timeit(lambda: tall_gemm(x, w, b)), timeit(lambda: F.linear(x, w, b))
        err = (tall_gemm(x, w, b).float() - F.linear(x.float(), w.float(), b.float())).abs().max().item()
        mb = (M * K + M * N + N * K) * 2 / 1e6
        print(f"M {M:6d} K {K:4d} N {N:4d}: native {t0:6.1f} us  library {t1:6.1f} us   stream floor {mb / 5.5e3 * 1e3:5.1f} us  max err {err:.3f}")
        if N != 96:
            tot[0] += t0 * 2; tot[1] += t1 * 2
print(f"K and V, forward + input gradient, three levels: native {tot[0]:.0f} us, library {tot[1]:.0f} us")
