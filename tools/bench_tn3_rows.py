#!/usr/bin/env python3
"""gemm3_tn3_kernel on 192-row against 176-row tiles (M = 43 008: 224 against 245 tiles on 256 CUs), interleaved rounds, HIP events."""
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
    for (N, K, addend) in ((256, 32, False), (256, 64, False), (256, 128, False), (256, 256, False), (256, 256, True), (256, 512, False), (256, 1024, False), (256, 288, False), (512, 512, False)):
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        cin = torch.randn(M, N, device=dev) if addend else None
        (pl, wam), = split_weights_grouped_h2([([w], False)])
        am = amax(a)
        oam = amax_slots(1, dev)
        outs = {}
        for rnd in range(4):
            for name, v in (("192", 0), ("176", 1)):
                _lib.set_option("gemm3_tn3_176", v)
                fn = lambda: gemm3_h2(a, am, pl, wam, b, cin=cin, out_amax=oam[0])  # noqa: E731
                outs[name] = fn()
                kern = _lib.last_kernel()
                t = timeit(fn)
                if rnd:
                    res.setdefault((N, K, addend, name, kern), []).append(t)
        assert torch.equal(outs["192"], outs["176"]), (N, K)
    _lib.set_option("gemm3_tn3_176", 1)
    # box calibration: kernels this experiment does not touch
    a = torch.randn(M, 256, device=dev); w = torch.randn(1024, 256, device=dev) / 16; b = torch.randn(1024, device=dev)
    (pl, wam), = split_weights_grouped_h2([([w], False)]); am = amax(a)
    print(json.dumps({"calibration": "gemm3_ws N=1024 K=256", "us": round(timeit(lambda: gemm3_h2(a, am, pl, wam, b)), 1), "kernel": _lib.last_kernel()}))
    x = torch.empty(55 * 1000 * 1000, device=dev); y = torch.empty_like(x)
    print(json.dumps({"calibration": "torch copy 220 MB", "us": round(timeit(lambda: y.copy_(x)), 1)}))
    _lib.set_option("gemm3_tn3", 0)
    a = torch.randn(M, 1024, device=dev); w = torch.randn(256, 1024, device=dev) / 32; b = torch.randn(256, device=dev)
    (pl, wam), = split_weights_grouped_h2([([w], False)]); am = amax(a)
    print(json.dumps({"calibration": "two-pass kernel N=256 K=1024", "us": round(timeit(lambda: gemm3_h2(a, am, pl, wam, b)), 1), "kernel": _lib.last_kernel()}))
    _lib.set_option("gemm3_tn3", 1)
    for (N, K, addend, name, kern), ts in res.items():
        ts = sorted(ts)
        print(json.dumps({"N": N, "K": K, "addend": addend, "rows": name, "kernel": kern, "us_med": round(ts[len(ts) // 2], 1), "us_min": round(ts[0], 1)}))


if __name__ == "__main__":
    main()
