#!/usr/bin/env python3
"""Row-local chains of the decoder's query side (csrc/row_chain.hip) against the launches they replace, in-stream back to back."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mp_former_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
st = _lib.stream_ptr(dev)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 600
a = torch.randn(rows, 256, device=dev).bfloat16()
ws = [(torch.randn(256, 256, device=dev) / 16).bfloat16() for _ in range(3)]
bs = [torch.randn(256, device=dev).bfloat16() for _ in range(3)]
x = torch.randn(rows, 256, device=dev)
gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
s, y32 = torch.empty_like(x), torch.empty_like(x)
y16, t, e1, e2 = (torch.empty(rows, 256, device=dev, dtype=torch.bfloat16) for _ in range(4))
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)


def chain_ln():
    lib.mpf_lin256_res_ln_forward(a.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), s.data_ptr(),
                                  y32.data_ptr(), y16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, 1e-5, st)


def sep_ln():
    lib.mpf_small_gemm_bf16(a.data_ptr(), 256, 1, None, ws[0].data_ptr(), 256, 1, bs[0].data_ptr(), None, 0, t.data_ptr(), 256, None, rows, 256, 256, 0, st)
    lib.mpf_res_ln256_forward(x.data_ptr(), t.data_ptr(), _lib.MPF_BF16, gamma.data_ptr(), beta.data_ptr(), s.data_ptr(), y32.data_ptr(), y16.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), rows, 1e-5, None, 0, None, st)


def chain_mlp():
    lib.mpf_ln256_mlp3_forward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(),
                               ws[2].data_ptr(), bs[2].data_ptr(), y16.data_ptr(), rows, 1e-5, st)


def sep_mlp():
    lib.mpf_res_ln256_forward(x.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(), None, None, y16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows,
                              1e-5, None, 0, None, st)
    for src, dst, k, relu in ((y16, e1, 0, 1), (e1, e2, 1, 1), (e2, y16, 2, 0)):
        lib.mpf_small_gemm_bf16(src.data_ptr(), 256, 1, None, ws[k].data_ptr(), 256, 1, bs[k].data_ptr(), None, 0, dst.data_ptr(), 256, None, rows, 256,
                                256, relu, st)


for name, fn in (("out-proj + residual + LN: chain", chain_ln), ("out-proj + residual + LN: 2 launches", sep_ln),
                 ("decoder_norm + mask_embed: chain", chain_mlp), ("decoder_norm + mask_embed: 4 launches", sep_mlp)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n = 300
    e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1_.record()
    torch.cuda.synchronize()
    print(f"{name:42s} rows {rows}: {e0.elapsed_time(e1_) * 1000 / n:6.2f} us per call (in-stream, back to back)")
