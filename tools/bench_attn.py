#!/usr/bin/env python3
"""Attention core kernels alone (csrc/attn.hip) at the decoder's shapes: forward (+ combine) and backward (dK/dV kernel, dQ kernel
+ sum) per level, with and without the byte mask.  usage: bench_attn.py [Lq] [N]   (MPF_OPTIONS=attn_kpw=.. for the split sweep)"""
import math
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mp_former_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 120
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H, E = 8, 256
lib = _lib.lib()
st = torch.cuda.current_stream(dev).cuda_stream


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for Lk in (1024, 4096, 16384, Lq):
    torch.manual_seed(0)
    q = torch.randn(Lq, N, E, device=dev).bfloat16()
    k = torch.randn(Lk, N, E, device=dev).bfloat16()
    v = torch.randn(Lk, N, E, device=dev).bfloat16()
    do = torch.randn(Lq, N, E, device=dev).bfloat16()
    kt = k.permute(1, 2, 0).contiguous()
    vt = v.permute(1, 2, 0).contiguous()
    LqP = (Lq + 31) // 32 * 32
    qT = torch.zeros(N, E, LqP, device=dev, dtype=torch.bfloat16); qT[:, :, :Lq] = q.permute(1, 2, 0)
    doT = torch.zeros(N, E, LqP, device=dev, dtype=torch.bfloat16); doT[:, :, :Lq] = do.permute(1, 2, 0)
    mask = (torch.rand(N, Lq, Lk, device=dev) < 0.5)
    mask[:, :, 0] = False
    out = torch.empty(Lq, N, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(N, H, Lq, device=dev)
    delta = torch.zeros(N, H, Lq, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(lib.mpf_attn_workspace_bytes(Lq, Lk, N, H) + 1024, dtype=torch.uint8, device=dev)
    sc = 1.0 / math.sqrt(32)
    res = {}
    for name, m in (("mask", mask), ("nomask", None)):
        mp = m.data_ptr() if m is not None else None

        def fwd():
            _lib.check(lib.mpf_attn_forward(q.data_ptr(), k.data_ptr(), vt.data_ptr(), mp, 1, out.data_ptr(), lse.data_ptr(), Lq, Lk, N, H, 32,
                                            sc, ws.data_ptr(), ws.numel(), st), "fwd")

        aux = torch.empty(lib.mpf_attn_bwd_aux_bytes(Lq, Lk, N, H, N) + 16, dtype=torch.uint8, device=dev)

        def bwd():          # as mpf_decoder_layer_backward issues it: prep (+ aux) launch, dK / dV kernel, dQ kernel (+ sum)
            _lib.check(lib.mpf_attn_bwd_prep_aux(q.data_ptr(), do.data_ptr(), out.data_ptr(), lse.data_ptr(), mp, 1, Lk, qT.data_ptr(),
                                                 doT.data_ptr(), delta.data_ptr(), aux.data_ptr(), aux.numel(), Lq, LqP, N, H, st), "prep")
            _lib.check(lib.mpf_attn_backward_kv_aux(q.data_ptr(), k.data_ptr(), v.data_ptr(), 0, 0, kt.data_ptr(), qT.data_ptr(),
                                                    do.data_ptr(), doT.data_ptr(), mp, 1, lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                                                    dk.data_ptr(), dv.data_ptr(), 0, 0, Lq, LqP, Lk, N, H, 32, sc, ws.data_ptr(),
                                                    ws.numel(), aux.data_ptr(), st), "bwd")
        fwd()
        res[name] = (timeit(fwd), timeit(bwd))
    print(f"Lq {Lq} Lk {Lk:6d} N {N}: forward {res['mask'][0]:6.1f} us (no mask {res['nomask'][0]:6.1f})   "
          f"backward {res['mask'][1]:6.1f} us (no mask {res['nomask'][1]:6.1f})", flush=True)
