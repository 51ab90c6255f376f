#!/usr/bin/env python3
"""One variant of the encoder FFN products at config B (R = 43 008), a few launches, for the PMC passes of tools/pmc_gate_bits.sh:
PROBE_GATE = 0 linear1 + ReLU | 1 the same with the gate-mask output | 2 dh with the activation as gate | 3 dh with the bit mask"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mp_former_amd.gemm3 import amax, gemm3_h2, gemm3_h2_bits, split_weights_grouped_h2  # noqa: E402

dev = torch.device("cuda:0")
which = int(os.environ.get("PROBE_GATE", "0"))
M, C, F = 43008, 256, 1024
torch.manual_seed(0)
x, g = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
w1, w2 = torch.randn(F, C, device=dev) / 16, torch.randn(C, F, device=dev) / 32
(p1, a1), (p2t, a2t) = split_weights_grouped_h2([([w1], False), ([w2], True)])
xa, ga = amax(x), amax(g)
h, bits = gemm3_h2_bits(x, xa, p1, a1, relu=True, want_bits=True)
torch.cuda.synchronize()
fn = [lambda: gemm3_h2(x, xa, p1, a1, relu=True), lambda: gemm3_h2_bits(x, xa, p1, a1, relu=True, want_bits=True),
      lambda: gemm3_h2(g, ga, p2t, a2t, gate=h), lambda: gemm3_h2_bits(g, ga, p2t, a2t, gate_bits=bits)][which]
for _ in range(12):
    fn()
torch.cuda.synchronize()
