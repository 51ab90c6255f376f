#!/usr/bin/env python3
"""Weight gradient of the FPN's 3x3 convolution (256 -> 256 channels at 256 x 256, N = 2): 256 x 256 tiles (csrc/gemm3_nt2.h) against
the 128 x 128 tiles (`gemm3_nt2=0`): time and agreement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib, conv3x3  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(2, 256, 256, 256, device=dev).permute(0, 3, 1, 2)      # channel-last planes as an [N, C, H, W] view
x.requires_grad_(True)
w = (torch.randn(256, 256, 3, 3, device=dev) / 48).requires_grad_(True)
b = torch.zeros(256, device=dev, requires_grad=True)
g = torch.randn(2, 256, 256, 256, device=dev).permute(0, 3, 1, 2)
ref = None
for on in (0, 1):
    _lib.set_option("gemm3_nt2", on)
    for _ in range(2):
        conv3x3.conv3x3(x, w, b).backward(g)
    _lib.profile_enable(True)
    for _ in range(5):
        w.grad = None
        conv3x3.conv3x3(x, w, b).backward(g)
    torch.cuda.synchronize()
    n, ms, _ = _lib.profile_get("gemm3_nt_kernel<conv3x3")
    _lib.profile_enable(False)
    print(f"gemm3_nt2={on}: conv wgrad {ms / max(n, 1) * 1e3:.1f} us ({n} launches) [{_lib.last_kernel()}]", flush=True)
    if ref is None:
        ref = w.grad.clone()
print("max |dW difference| / max |dW|:", float((w.grad - ref).abs().max() / ref.abs().max()))
_lib.set_option("gemm3_nt2", 1)
