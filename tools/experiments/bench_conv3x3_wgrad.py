"""Weight gradient of the 3x3 FPN convolution: MIOpen's fp32 wrw against the split-bf16 NT kernel at the equivalent plain
shape (R = N*H*W rows, [Cout] x [9*Cin] output; no tap shifts — an upper bound for a convolution mode of that kernel)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from tools.bench_gemm3 import timeit
from mp_former_amd.gemm3 import gemm3_nt, nt_reduce
dev = torch.device("cuda:0")
torch.manual_seed(0)
R = 2 * 256 * 256
g = torch.randn(R, 256, device=dev); x9 = torch.randn(R, 2304, device=dev)
for rps in (9376, 4704, 2368):
    ts = sorted(timeit(lambda: nt_reduce(*gemm3_nt(g, x9, rps, want_csum_a=False)[:1]), 5) for _ in range(3))
    print(f"gemm3_nt R={R} 256x2304 rps={rps}: {ts[1]:.1f} us  {2.0*R*256*2304/ts[1]/1e6:.0f} TF")
x = torch.randn(2, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
gy = torch.randn(2, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 3, 3, device=dev)
f = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
ts = sorted(timeit(f, 5) for _ in range(3))
print(f"MIOpen wrw: {ts[1]:.1f} us  {2.0*R*256*2304/ts[1]/1e6:.0f} TF")
