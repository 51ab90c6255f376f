#!/usr/bin/env python3
"""Accuracy (vs fp64) and speed (vs torch fp32 addmm) of the split-bf16 fp32 GEMM on the encoder shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib  # noqa: E402


def stream():
    return torch.cuda.current_stream().cuda_stream


def split(w, transpose=False):
    r, c = w.shape
    out = torch.empty((3, c, r) if transpose else (3, r, c), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().mpf_gemm3_split(w.data_ptr(), r, c, 1 if transpose else 0, out.data_ptr(), stream()), "split")
    return out


def gemm3(a, planes, bias=None, a2=None, cin=None, relu=False, out=None):
    M, K = a.shape
    N = planes.shape[1]
    c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().mpf_gemm3_tn(
        a.data_ptr(), a.stride(0), a2.data_ptr() if a2 is not None else None, a2.shape[0] if a2 is not None else 0,
        planes.data_ptr(), bias.data_ptr() if bias is not None else None,
        cin.data_ptr() if cin is not None else None, cin.stride(0) if cin is not None else 0,
        None, 0, None, 0, c.data_ptr(), c.stride(0), M, N, K, 1 if relu else 0, stream()), "gemm3")
    return c


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
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M = 43008
    # correctness on a ragged problem first
    for (m, n, k) in ((300, 288, 64), (129, 100, 32), (128, 256, 256)):
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev)
        b = torch.randn(n, device=dev)
        a2 = torch.randn(7, k, device=dev)
        cin = torch.randn(m, n, device=dev)
        ref = ((a.double() + a2.double()[torch.arange(m, device=dev) % 7]) @ w.double().t() + b.double() + cin.double()).relu()
        got = gemm3(a, split(w), b, a2=a2, cin=cin, relu=True)
        err = (got.double() - ref).abs().max().item()
        lib = (torch.addmm(b, a + a2[torch.arange(m, device=dev) % 7], w.t()) + cin).relu()
        print(f"ragged {m}x{n}x{k}: max err gemm3 {err:.3e}  torch fp32 {(lib.double() - ref).abs().max().item():.3e}")
        wt = split(w, transpose=True)     # [3, k, n]: planes of w^T
        g = torch.randn(m, n, device=dev)
        if n % 32 == 0:
            dx = gemm3(g, wt)
            refdx = g.double() @ w.double()
            print(f"   dX: err {(dx.double() - refdx).abs().max().item():.3e}  torch {((g @ w).double() - refdx).abs().max().item():.3e}")
    for (n, k, name) in ((256, 256, "value/output proj"), (288, 256, "offsets+weights"), (1024, 256, "ffn1"),
                         (256, 1024, "ffn2")):
        a = torch.randn(M, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        planes = split(w)
        ref = a.double() @ w.double().t() + b.double()
        got = gemm3(a, planes, b)
        lib = torch.addmm(b, a, w.t())
        e3 = (got.double() - ref).abs()
        el = (lib.double() - ref).abs()
        t3 = timeit(lambda: gemm3(a, planes, b))
        tl = timeit(lambda: torch.addmm(b, a, w.t()))
        fl = 2.0 * M * n * k
        print(f"{name:18s} N={n:5d} K={k:5d}: gemm3 {t3:7.1f} us ({fl / t3 / 1e6:6.1f} TF)  torch {tl:7.1f} us ({fl / tl / 1e6:6.1f} TF)"
              f"  err max/mean gemm3 {e3.max().item():.2e}/{e3.mean().item():.2e} torch {el.max().item():.2e}/{el.mean().item():.2e}",
              flush=True)


if __name__ == "__main__":
    main()
