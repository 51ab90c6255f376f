#!/usr/bin/env python3
"""FFN pair of an encoder layer at config B (R = 43 008, 256 -> 1024 -> 256): linear1 + ReLU with / without the gate-mask output,
and the gated input-gradient product with the activation as gate (reads 4 bytes per element) against the bit mask."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mp_former_amd.gemm3 import amax, gemm3_h2, gemm3_h2_bits, split_weights_grouped_h2  # noqa: E402

dev = torch.device("cuda:0")
M, C, F = 43008, 256, 1024
torch.manual_seed(0)
x, g = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
w1, w2 = torch.randn(F, C, device=dev) / 16, torch.randn(C, F, device=dev) / 32
(p1, a1), (p2t, a2t) = split_weights_grouped_h2([([w1], False), ([w2], True)])
xa, ga = amax(x), amax(g)


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


h, bits = gemm3_h2_bits(x, xa, p1, a1, relu=True, want_bits=True)
for rep in range(2):
    print("linear1 + relu            %6.1f us" % timeit(lambda: gemm3_h2(x, xa, p1, a1, relu=True)))
    print("linear1 + relu + mask out %6.1f us" % timeit(lambda: gemm3_h2_bits(x, xa, p1, a1, relu=True, want_bits=True)))
    print("dh, activation gate       %6.1f us" % timeit(lambda: gemm3_h2(g, ga, p2t, a2t, gate=h)))
    print("dh, bit-mask gate         %6.1f us" % timeit(lambda: gemm3_h2_bits(g, ga, p2t, a2t, gate_bits=bits)))
