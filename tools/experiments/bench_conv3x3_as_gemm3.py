import sys, torch
sys.path.insert(0, "/root/repo")
from tools.bench_gemm3 import gemm3, split, timeit
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 2 * 258 * 258
for (n, k) in ((256, 2304), (256, 256)):
    a = torch.randn(M, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    pw = split(w)
    ts = sorted(timeit(lambda: gemm3(a, pw, b)) for _ in range(3))
    print(f"M={M} N={n} K={k}: {ts[1]:.1f} us  {2.0*M*n*k/ts[1]/1e6:.0f} TF")
# MIOpen 3x3 fp32 conv for comparison
x = torch.randn(2, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
cw = torch.randn(256, 256, 3, 3, device=dev)
import torch.nn.functional as F
ts = sorted(timeit(lambda: F.conv2d(x, cw, None, 1, 1)) for _ in range(3))
print(f"MIOpen conv3x3 fwd: {ts[1]:.1f} us  {2.0*2*65536*256*256*9/ts[1]/1e6:.0f} TF")
