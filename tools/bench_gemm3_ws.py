#!/usr/bin/env python3
"""fp16 x 2 TN products of the encoder at M = 43 008 (config B, N = 2): the weight-stationary kernel (csrc/gemm3_ws.h) against the
tiled kernels (`gemm3_ws=0`), per shape and epilogue, alone on the GPU: time, algorithmic GB/s, MFMA TFLOP/s issued.
    python tools/bench_gemm3_ws.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib  # noqa: E402
from mp_former_amd.gemm3 import amax, amax_slots, gemm3_h2, gemm3_h2_bits, split_weights_grouped_h2  # noqa: E402


def timeit(fn, iters=30):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 43008
    torch.manual_seed(0)
    # rotate over several input buffers so that no operand is L2 / MALL resident from the previous call (as in the step)
    NB = 6
    for (N, K, name, opts) in ((256, 256, "value_proj", {}), (256, 256, "output_proj + residual", {"cin": 1}),
                               (1024, 256, "linear1 + relu + bits", {"relu": 1, "bits": 1}), (1024, 256, "dh (bit gate)", {"gbits": 1}),
                               (256, 1024, "linear2 + residual (tiled)", {"cin": 1}), (288, 256, "offsets+weights (tiled)", {})):
        As = [torch.randn(M, K, device=dev) for _ in range(NB)]
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        cin = torch.randn(M, N, device=dev) if opts.get("cin") else None
        (pl, wam), = split_weights_grouped_h2([([w], False)])
        ams = [amax(a) for a in As]
        oam = amax_slots(1, dev)[0]
        bits = torch.randint(0, 256, (M, N // 8), dtype=torch.uint8, device=dev) if opts.get("gbits") else None
        it = [0]

        def run():
            i = it[0] = (it[0] + 1) % NB
            if opts.get("bits"):
                return gemm3_h2_bits(As[i], ams[i], pl, wam, b, relu=True, out_amax=oam, want_bits=True)
            if bits is not None:
                return gemm3_h2_bits(As[i], ams[i], pl, wam, gate_bits=bits, out_amax=oam)
            return gemm3_h2(As[i], ams[i], pl, wam, b, cin=cin, relu=bool(opts.get("relu")))
        out = []
        for ws in (0, 1):
            _lib.set_option("gemm3_ws", 256 if ws else 0)
            t = timeit(run)
            kern = _lib.last_kernel()
            by = 4.0 * (M * K + M * N + (M * N if cin is not None else 0)) + 4.0 * N * K
            out.append(f"ws={ws}: {t:6.1f} us  {by / t / 1e3:6.0f} GB/s  {6.0 * M * N * K / t / 1e6:6.0f} TF  [{kern}]")
        _lib.set_option("gemm3_ws", 512)
        print(f"{name:28s} N={N:5d} K={K:5d} | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
