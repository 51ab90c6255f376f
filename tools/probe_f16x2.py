#!/usr/bin/env python3
"""The fp16 x 2 form of gemm3 (two pieces per operand, three products) next to the six-product bf16 x 3 form and the library
fp32 GEMM on the encoder's TN shapes: error against fp64 (relative to sum |a||b|) and time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd.gemm3 import amax, amax_slots, amax_value, gemm3, gemm3_h2, split_weight, split_weights_grouped_h2  # noqa: E402
from tools.bench_gemm3 import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 43008
SHAPES = ((256, 256), (1024, 256), (256, 1024))
if os.environ.get("PROBE_SHAPE"):            # one shape only (tools/pmc_gemm3_traffic.sh)
    SHAPES = (SHAPES[int(os.environ["PROBE_SHAPE"])],)
for (n, k) in SHAPES:
    a = torch.randn(M, k, device=dev) * (1 + 9 * torch.rand(M, 1, device=dev))
    a[::7, ::13] *= 30
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    ref = a.double() @ w.double().t() + b.double()
    den = a.double().abs() @ w.double().abs().t() + b.double().abs()
    p3 = split_weight(w)
    got6 = gemm3(a, p3, b)
    t6 = timeit(lambda: gemm3(a, p3, b))
    (p2, wam), (p2t, wamt) = split_weights_grouped_h2([([w], False), ([w], True)])
    am = amax(a)
    oam = amax_slots(1, dev)[0]
    got3 = gemm3_h2(a, am, p2, wam, b, out_amax=oam)
    assert float(amax_value(am)) == float(a.abs().max()) and float(amax_value(wam)) == float(w.abs().max())
    assert float(amax_value(oam)) == float(got3.abs().max())
    t3 = timeit(lambda: gemm3_h2(a, am, p2, wam, b))
    t3o = timeit(lambda: gemm3_h2(a, am, p2, wam, b, out_amax=oam))
    tam = timeit(lambda: amax(a))
    lib = torch.addmm(b, a, w.t())

    def e(x):
        d = (x.double() - ref).abs() / den
        return f"max {float(d.max()):.2e} mean {float(d.mean()):.2e}"
    print(f"N={n} K={k}: bf16x3/6 {t6:6.1f} us [{e(got6)}]   fp16x2/3 {t3:6.1f} us (with out_amax {t3o:.1f}; amax pass {tam:.1f}) [{e(got3)}]   library fp32 [{e(lib)}]", flush=True)
    # the transposed planes: dX = g . W
    g = torch.randn(M, n, device=dev)
    if k % 256 == 0:
        dx = gemm3_h2(g, amax(g), p2t, wamt)
        refdx = g.double() @ w.double()
        print(f"   dX: max err {(dx.double() - refdx).abs().max().item():.3e}  library {((g @ w).double() - refdx).abs().max().item():.3e}")
