#!/usr/bin/env python3
"""Probe: what would a grouped weight-gradient launch buy?  The same NT kernel on k times the rows with k times the
rows per split has the workgroup count of ONE problem and the K-loop length / partial-sum traffic per problem of a
k-problem group."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd.gemm3 import gemm3_nt, nt_reduce  # noqa: E402
from tools.bench_gemm3 import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    R0 = 43008
    for (M, N) in ((256, 256), (1024, 256), (256, 1024)):
        tiles = (M // 128) * (N // 128)
        for k in (1, 2, 3, 5, 10):
            R = R0 * k
            a = torch.randn(R, M, device=dev)
            b = torch.randn(R, N, device=dev)
            ns = max(1, 512 // tiles)
            rps = ((R + ns - 1) // ns + 31) // 32 * 32
            t = timeit(lambda: gemm3_nt(a, b, rps, want_csum_a=True))
            c, ca, _ = gemm3_nt(a, b, rps, want_csum_a=True)
            tr = timeit(lambda: nt_reduce(c, ca))
            print(f"M={M} N={N} k={k}: rps {rps} nsplit {c.shape[0]}  nt {t:7.1f} us = {t / k:6.1f} per problem "
                  f"({2.0 * R * M * N / t / 1e6:6.1f} TF)  reduce {tr:5.1f} us", flush=True)
            del a, b, c


if __name__ == "__main__":
    main()
