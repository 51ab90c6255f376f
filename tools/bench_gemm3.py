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


def main_nt():
    from mp_former_amd.gemm3 import gemm3_nt, pick_rows_per_split
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (R, m, n, rps, b2r) in ((300, 288, 64, 64, 40), (129, 100, 36, 32, None), (672, 256, 256, 96, 336)):
        a = torch.randn(R, m, device=dev)
        b = torch.randn(R, n, device=dev)
        b2 = torch.randn(b2r, n, device=dev) if b2r else None
        bb = b.double() + (b2.double()[torch.arange(R, device=dev) % b2r] if b2r else 0)
        ref = a.double().t() @ bb
        for tr in (False, True):
            c, ca, cb = gemm3_nt(a, b, rps, b2=b2, want_csum_a=True, want_csum_b=True, transpose_out=tr)
            got = c.sum(0).double()
            got = got.t() if tr else got
            print(f"nt {R}x{m}x{n} tr={tr}: err {(got - ref).abs().max().item():.3e} torch {((a.t() @ bb.float()).double() - ref).abs().max().item():.3e}"
                  f"  csum_a {(ca.sum(0).double() - a.double().sum(0)).abs().max().item():.2e} csum_b {(cb.sum(0).double() - bb.sum(0)).abs().max().item():.2e}")
    R = 43008
    for (m, n, name) in ((256, 256, "dW value/out"), (256, 288, "dW offsets (q x draw)"), (1024, 256, "dW ffn1"), (256, 1024, "dW ffn2")):
        a = torch.randn(R, m, device=dev)
        b = torch.randn(R, n, device=dev)
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        rps = pick_rows_per_split(R, tiles, 1024)
        ref = a.double().t() @ b.double()
        c, _, cb = gemm3_nt(a, b, rps, want_csum_b=True)
        e3 = (c.sum(0).double() - ref).abs()
        chunks = 32
        lib = torch.bmm(a.view(chunks, R // chunks, -1).transpose(1, 2), b.view(chunks, R // chunks, -1)).sum(0)
        el = (lib.double() - ref).abs()
        t3 = timeit(lambda: (gemm3_nt(a, b, rps, want_csum_b=True)[0].sum(0)))
        tl = timeit(lambda: (torch.bmm(a.view(chunks, R // chunks, -1).transpose(1, 2), b.view(chunks, R // chunks, -1)).sum(0), b.sum(0)))
        fl = 2.0 * R * m * n
        print(f"{name:22s} {m}x{n} rps={rps}: gemm3_nt+sum {t3:7.1f} us ({fl / t3 / 1e6:6.1f} TF)  bmm+sums {tl:7.1f} us ({fl / tl / 1e6:6.1f} TF)"
              f"  err gemm3 {e3.max().item():.2e} torch {el.max().item():.2e}", flush=True)


def main_ablate():
    dev = torch.device("cuda:0")
    M = 43008
    for (n, k) in ((256, 256), (256, 1024), (1024, 256)):
        a = torch.randn(M, k, device=dev)
        w = torch.randn(n, k, device=dev)
        planes = split(w)
        out = []
        for ab in (0, 1, 2, 4, 5, 6, 7):
            _lib.set_option("gemm3_ablate", ab)
            out.append(f"ab{ab}:{timeit(lambda: gemm3(a, planes)):6.1f}")
        _lib.set_option("gemm3_ablate", 0)
        print(f"N={n} K={k}: " + "  ".join(out) + " us", flush=True)


def main_rps():
    from mp_former_amd.gemm3 import gemm3_nt
    dev = torch.device("cuda:0")
    R = 43008
    pos = torch.randn(21504, 256, device=dev)
    for (m, n, name, b2) in ((256, 256, "dW value/out", None), (288, 256, "dW288 (draw x q, b2)", pos), (1024, 256, "dW ffn1", None),
                             (256, 1024, "dW ffn2", None)):
        a = torch.randn(R, m, device=dev)
        b = torch.randn(R, n, device=dev)
        out = []
        for rps in (256, 512, 1024, 2048):
            t3 = timeit(lambda: (gemm3_nt(a, b, rps, b2=b2, want_csum_a=True)[0].sum(0)))
            out.append(f"rps{rps}:{t3:6.1f}")
        print(f"{name:22s} {m}x{n}: " + "  ".join(out) + " us", flush=True)


def main_l2():
    """A operand with row stride 0 (every row the same 1-4 KB: L1/L2 resident) vs the real matrix."""
    dev = torch.device("cuda:0")
    M = 43008
    for (n, k) in ((256, 256), (256, 1024), (1024, 256)):
        a = torch.randn(M, k, device=dev)
        a0 = torch.randn(1, k, device=dev).expand(M, k)
        planes = split(torch.randn(n, k, device=dev))
        c = torch.empty(M, n, device=dev)
        t_real = timeit(lambda: gemm3(a, planes, out=c))
        t_l2 = timeit(lambda: gemm3(a0, planes, out=c))
        print(f"N={n} K={k}: real A {t_real:.1f} us, cache-resident A {t_l2:.1f} us", flush=True)


def main_phases():
    """needs a library built with -DG3_TIMING (make -C mp_former_amd/csrc clean all CXXFLAGS='... -DG3_TIMING')"""
    import ctypes
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    fn = lib.mpf_gemm3_debug_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    M = 43008
    for (n, k) in ((256, 256), (256, 1024), (1024, 256)):
        a = torch.randn(M, k, device=dev)
        planes = split(torch.randn(n, k, device=dev))
        gemm3(a, planes); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        fn(buf, 1)
        t = timeit(lambda: gemm3(a, planes), iters=10)
        torch.cuda.synchronize()
        fn(buf, 1)
        steps = buf[5]
        names = ["barrier1", "split+write", "barrier2", "load issue", "mma step"]
        per = [buf[i] / steps for i in range(5)]
        print(f"N={n} K={k}: {t:.1f} us; per K-step of wave 0 (s_memtime ticks @100MHz?): " +
              ", ".join(f"{nm} {v:.1f}" for nm, v in zip(names, per)) + f"  total {sum(per):.1f}", flush=True)


def main_one():
    from mp_former_amd.gemm3 import gemm3_nt
    dev = torch.device("cuda:0")
    M = 43008
    a = torch.randn(M, 1024, device=dev)
    w = torch.randn(256, 1024, device=dev)
    planes = split(w)
    b = torch.randn(M, 256, device=dev)
    for _ in range(3):
        gemm3(a, planes)
        gemm3_nt(a, b, 1024, want_csum_b=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if "--l2" in sys.argv:
        main_l2()
        sys.exit(0)
    if "--phases" in sys.argv:
        main_phases()
        sys.exit(0)
    if "--rps" in sys.argv:
        main_rps()
        sys.exit(0)
    if "--one" in sys.argv:
        main_one()
        sys.exit(0)
    if "--ablate" in sys.argv:
        main_ablate()
        sys.exit(0)
    if "--nt" in sys.argv:
        main_nt()
        sys.exit(0)
    main()
