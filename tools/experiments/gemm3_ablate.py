"""Timing of the gemm3 TN kernel built with -DG3_ABL=<bits> (1 no loads in the loop, 2 no split arithmetic, 4 no LDS writes,
8 no barriers): MPF_LIB_PATH=ab/lib_abl<bits>.so python tools/experiments/gemm3_ablate.py — results are WRONG by construction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench_gemm3 import gemm3, split, timeit
dev = torch.device("cuda:0")
M = 43008
out = []
for (n, k) in ((256, 256), (256, 1024), (1024, 256)):
    a = torch.randn(M, k, device=dev)
    planes = split(torch.randn(n, k, device=dev))
    c = torch.empty(M, n, device=dev)
    out.append(f"N={n} K={k}: {timeit(lambda: gemm3(a, planes, out=c), 30):6.1f} us")
print(os.environ.get("MPF_LIB_PATH", "default"), " | ".join(out))
