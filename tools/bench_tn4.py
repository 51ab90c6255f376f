#!/usr/bin/env python3
"""gemm3_tn4_kernel (192 x 256 tiles, row halves half a K step apart) against gemm3_tn3_kernel (lockstep) on the encoder's TN
shapes, M = 43 008: interleaved rounds, HIP events, median.   python tools/bench_tn4.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib  # noqa: E402
from mp_former_amd.gemm3 import amax, amax_slots, gemm3_h2, split_weights_grouped_h2  # noqa: E402


def timeit(fn, iters=20):
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


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M = int(os.environ.get("M", "43008"))
    res = {}
    for (N, K, addend) in ((256, 256, False), (256, 256, True), (256, 1024, False), (1024, 256, False), (512, 512, False)):
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        cin = torch.randn(M, N, device=dev) if addend else None
        (pl, wam), = split_weights_grouped_h2([([w], False)])
        am = amax(a)
        oam = amax_slots(1, dev)
        outs = {}
        for rnd in range(4):
            for name, opts in (("tn3", dict(gemm3_tn4=0, gemm3_ws=0)), ("tn4", dict(gemm3_tn4=1, gemm3_ws=0)), ("default", dict(gemm3_tn4=1, gemm3_ws=512))):
                for k_, v_ in opts.items():
                    _lib.set_option(k_, v_)
                fn = lambda: gemm3_h2(a, am, pl, wam, b, cin=cin, out_amax=oam[0])  # noqa: E731
                outs[name] = fn()
                kern = _lib.last_kernel()
                t = timeit(fn)
                if rnd:
                    res.setdefault((N, K, addend, name, kern), []).append(t)
        assert torch.equal(outs["tn3"], outs["tn4"]), (N, K)
    _lib.set_option("gemm3_tn4", 1)
    _lib.set_option("gemm3_ws", 512)
    for (N, K, addend, name, kern), ts in res.items():
        ts = sorted(ts)
        by = 4.0 * (M * K + M * N * (2 if addend else 1)) + 4.0 * N * K
        print(json.dumps({"N": N, "K": K, "addend": addend, "route": name, "kernel": kern, "us_med": round(ts[len(ts) // 2], 1), "us_min": round(ts[0], 1),
                          "alg_TBps": round(by / ts[len(ts) // 2] / 1e6, 2), "issued_PF": round(6.0 * M * N * K / ts[len(ts) // 2] / 1e9, 3)}))


if __name__ == "__main__":
    main()
