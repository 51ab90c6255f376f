#!/usr/bin/env python3
"""Planar (pre-split operands, DMA-fed) fp32 GEMM vs the on-the-fly-split kernel on the encoder shapes:
accuracy vs fp64, bit-equality with the existing kernel, timings (GEMM alone and with the conversion pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib  # noqa: E402
from tools.bench_gemm3 import gemm3, split, stream, timeit  # noqa: E402


def to_planes(x, a2=None):
    r, c = x.shape
    out = torch.empty((3, c // 8, r, 8), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().mpf_gemm3_to_planes(x.data_ptr(), x.stride(0), a2.data_ptr() if a2 is not None else None,
                                              a2.shape[0] if a2 is not None else 0, out.data_ptr(), r, c, stream()), "to_planes")
    return out


def planar(ap, bp, M, N, K, bias=None, cin=None, relu=False, want_planes=False):
    c = torch.empty((M, N), dtype=torch.float32, device=ap.device)
    cp = torch.empty((3, N // 8, M, 8), dtype=torch.bfloat16, device=ap.device) if want_planes else None
    _lib.check(_lib.lib().mpf_gemm3_tn_planar(ap.data_ptr(), bp.data_ptr(), bias.data_ptr() if bias is not None else None,
                                              cin.data_ptr() if cin is not None else None, cin.stride(0) if cin is not None else 0,
                                              None, 0, None, 0, c.data_ptr(), c.stride(0), cp.data_ptr() if cp is not None else None,
                                              M, N, K, 1 if relu else 0, stream()), "planar")
    return c, cp


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (m, n, k) in ((300, 288, 64), (129, 104, 32), (256, 128, 256), (1000, 256, 1024)):
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        b = torch.randn(n, device=dev)
        cin = torch.randn(m, n, device=dev)
        ref = (a.double() @ w.double().t() + b.double() + cin.double()).relu()
        old = gemm3(a, split(w), b, cin=cin, relu=True)
        got, cp = planar(to_planes(a), to_planes(w), m, n, k, b, cin, True, want_planes=True)
        torch.cuda.synchronize()
        # planes of C must equal the planes of the fp32 C
        want_cp = to_planes(got)
        print(f"{m}x{n}x{k}: err planar {(got.double() - ref).abs().max().item():.3e} old {(old.double() - ref).abs().max().item():.3e} "
              f"bit-equal to old: {bool(torch.equal(got, old))}  C planes ok: {bool(torch.equal(cp, want_cp))}")
    M = 43008
    for (n, k, name) in ((256, 256, "value/output proj"), (1024, 256, "ffn1"), (256, 1024, "ffn2")):
        a = torch.randn(M, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        pw, bp = split(w), to_planes(w)
        ap = to_planes(a)
        res = {}
        for rnd in range(3):
            res.setdefault("old", []).append(timeit(lambda: gemm3(a, pw, b)))
            res.setdefault("planar", []).append(timeit(lambda: planar(ap, bp, M, n, k, b)))
            res.setdefault("planar+Cplanes", []).append(timeit(lambda: planar(ap, bp, M, n, k, b, want_planes=True)))
            res.setdefault("to_planes(A)", []).append(timeit(lambda: to_planes(a)))
        fl = 2.0 * M * n * k
        print(f"{name:20s} N={n} K={k}: " + "  ".join(f"{kk} {sorted(v)[1]:.1f} us ({fl / sorted(v)[1] / 1e6:.0f} TF)" if "to_" not in kk
                                                        else f"{kk} {sorted(v)[1]:.1f} us" for kk, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
