#!/usr/bin/env python3
"""1x1 convolutions of the R50 bottlenecks (bf16, channels_last, batch 2 at 1024^2), per pass: MIOpen (forward, input gradient,
weight gradient separately through aten.convolution_backward's output mask) vs the native bf16 GEMMs on the NHWC-flattened
activation (mpf_tall_gemm_bf16 forward / input gradient, mpf_gemm_nt_bf16 weight gradient)."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mp_former_amd import _miopen  # noqa: E402
from mp_former_amd.small_linear import gemm_nt_bf16, tall_gemm  # noqa: E402

dev = torch.device("cuda:0")
_miopen.use_shipped_find_db(check_version=False)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def flat(t):
    N, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(-1, C)


tot = {k: 0.0 for k in ("m_f", "m_dx", "m_dw", "n_f", "n_dx", "n_dw")}
for (cin, cout, hw, count) in ((64, 64, 256, 1), (64, 256, 256, 4), (256, 64, 256, 2), (256, 128, 256, 1), (128, 512, 128, 4),
                               (512, 128, 128, 3), (512, 256, 128, 1), (256, 1024, 64, 6), (1024, 256, 64, 5), (1024, 512, 64, 1),
                               (512, 2048, 32, 3), (2048, 512, 32, 2)):
    x = torch.randn(2, cin, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05
    g = torch.randn(2, cout, hw, hw, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x2, g2, w2 = flat(x), flat(g), w.view(cout, cin)
    wt = w2.t().contiguous()

    def cb(mask):
        return torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, mask)
    y0 = F.conv2d(x, w)
    y1 = tall_gemm(x2, w2)
    e_f = (flat(y0).float() - y1.float()).abs().max().item()
    dx0, dw0, _ = cb([True, True, False])
    dx1 = tall_gemm(g2, wt)
    dw1, _ = gemm_nt_bf16(g2, x2, want_csum=False)
    e_dx = (flat(dx0).float() - dx1.float()).abs().max().item() / flat(dx0).float().abs().max().item()
    e_dw = (dw0.view(cout, cin).float() - dw1.float()).abs().max().item() / dw0.float().abs().max().item()
    r = {"m_f": timeit(lambda: F.conv2d(x, w)), "m_dx": timeit(lambda: cb([True, False, False])),
         "m_dw": timeit(lambda: cb([False, True, False])), "n_f": timeit(lambda: tall_gemm(x2, w2)),
         "n_dx": timeit(lambda: tall_gemm(g2, wt)), "n_dw": timeit(lambda: gemm_nt_bf16(g2, x2, want_csum=False))}
    for k in tot:
        tot[k] += r[k] * count
    gb = 2 * 2 * hw * hw * (cin + cout) / 1e3     # KB moved by one pass, either direction
    print(f"{cin:5d}->{cout:5d} @{hw:3d}^2 x{count}: fwd {r['m_f']:6.1f} | {r['n_f']:6.1f}   dX {r['m_dx']:6.1f} | {r['n_dx']:6.1f}   "
          f"dW {r['m_dw']:6.1f} | {r['n_dw']:6.1f} us (MIOpen | native)  stream floor {gb / 5e3:5.1f} us   "
          f"err fwd {e_f:.3f} dx {e_dx:.4f} dw {e_dw:.4f}", flush=True)
print("sum over the network's stride-1 1x1 convs, ms: " + "  ".join(f"{k} {v / 1e3:.2f}" for k, v in tot.items()))
