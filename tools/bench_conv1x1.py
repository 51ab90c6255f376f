#!/usr/bin/env python3
"""1x1 convolutions of the R50 bottlenecks (bf16, channels_last, batch 2 at 1024^2): MIOpen conv2d vs
the same thing as a GEMM on the NHWC-flattened activation (forward + backward)."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def conv_mm(x, w):
    N, C, H, W = x.shape
    y = x.permute(0, 2, 3, 1).reshape(-1, C) @ w.view(w.shape[0], C).t()
    return y.view(N, H, W, -1).permute(0, 3, 1, 2)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot = {"miopen": 0.0, "mm": 0.0}
for (cin, cout, hw, count) in ((64, 64, 256, 1), (64, 256, 256, 4), (256, 64, 256, 2), (256, 128, 256, 1), (128, 512, 128, 4),
                               (512, 128, 128, 3), (512, 256, 128, 1), (256, 1024, 64, 6), (1024, 256, 64, 5), (1024, 512, 64, 1),
                               (512, 2048, 32, 3), (2048, 512, 32, 2)):
    x = torch.randn(2, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).requires_grad_(True)
    g = torch.randn(2, cout, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(f):
        y = f(x, w)
        y.backward(g)
        x.grad = None; w.grad = None
    y0, y1 = F.conv2d(x, w), conv_mm(x, w)
    err = (y0.float() - y1.float()).abs().max().item()
    t0 = timeit(lambda: run(lambda a, b: F.conv2d(a, b)))
    t1 = timeit(lambda: run(conv_mm))
    tot["miopen"] += t0 * count; tot["mm"] += t1 * count
    print(f"{cin:5d}->{cout:5d} @{hw:3d}^2 x{count}: conv2d fwd+bwd {t0:7.1f} us   mm fwd+bwd {t1:7.1f} us   max diff {err:.3f}", flush=True)
print(f"sum over the network's 1x1 convs: conv2d {tot['miopen'] / 1e3:.2f} ms, mm {tot['mm'] / 1e3:.2f} ms")
