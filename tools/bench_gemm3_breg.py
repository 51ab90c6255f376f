#!/usr/bin/env python3
"""fp16 x 2 two-pass TN tile: weight fragments from L2 into registers (gemm3_tn2r_kernel, mpf_set_option("gemm3_breg", 1)) vs
the LDS-staged form (0) on the encoder shapes at config B (M = 43 008): bit equality and time, interleaved rounds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib
from mp_former_amd.gemm3 import amax, amax_slots, gemm3_h2, split_weights_grouped_h2
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 43008


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (N, K) in ((256, 256), (1024, 256), (256, 1024), (512, 256)):
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    cin = torch.randn(M, N, device=dev)
    (planes, w_am), = split_weights_grouped_h2([([w], False)])
    a_am = amax(a)
    outs, times = {}, {0: [], 1: []}
    for rnd in range(3):
        for mode in (0, 1):
            _lib.set_option("gemm3_breg", mode)
            outs[mode] = gemm3_h2(a, a_am, planes, w_am, b, cin=cin, relu=True)
            kern = _lib.last_kernel()
            times[mode].append(timeit(lambda: gemm3_h2(a, a_am, planes, w_am, b, cin=cin, relu=True)))
    _lib.set_option("gemm3_breg", 1)
    eq = torch.equal(outs[0], outs[1])
    ref = (a.double() @ w.double().t() + b.double() + cin.double()).relu()
    err = float((outs[1].double() - ref).abs().max() / (ref.abs().max() + 1.0))
    t0, t1 = sorted(times[0])[1], sorted(times[1])[1]
    print(f"N={N:5d} K={K:5d}: LDS-staged {t0:7.1f} us, register-B {t1:7.1f} us ({t0 / t1:.2f}x)  bit-equal {eq}  max err vs fp64 {err:.2e}  [{kern}]")
