"""Masked multi-head attention core on the native bf16 MFMA kernel (csrc/attn.hip).

`attention_core(q, k, v, mask, nheads)`: q [Lq,N,E], k, v [Lk,N,E], mask bool [N,Lq,Lk] / [Lq,Lk] /
None (True = masked) -> [Lq,N,E]; the softmax(QK^T/sqrt(hd))V between the
packed in-projection and out_proj of nn.MultiheadAttention (decoder :42-52, :100-112).  bf16 I/O,
fp32 softmax and accumulation.  GPU only."""
import math

import torch
from torch.autograd import Function

from . import _lib

_ws = {}


def _workspace(device, nbytes):
    key = (device, _lib.ws_scope())          # (a graph capture has its own buffers: _lib.workspace_scope)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


class _AttnCore(Function):
    """q [Lq,N,E], k, v [Lk,N,E] (sequence-first, as nn.MultiheadAttention projects them), mask bool
    [N,Lq,Lk] / [Lq,Lk] / None -> [Lq,N,E].  Forward and backward on the native MFMA kernels; the
    transposed companions the kernels need (V^T, K^T, Q^T, dO^T) are plain copies made here."""

    @staticmethod
    def forward(ctx, q, k, v, mask, nheads):
        if not q.is_cuda:
            raise RuntimeError("mp_former_amd attention runs on the GPU only (no CPU fallback)")
        Lq, N, E = q.shape
        Lk = k.shape[0]
        hd = E // nheads
        if hd != 32:
            raise RuntimeError("native attention supports head_dim 32 (NHEADS 8 x HIDDEN_DIM 256)")
        qb, kb, vb = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
        lib = _lib.lib()
        stream = _lib.stream_ptr(q.device)
        kt = torch.empty((N, E, Lk), dtype=torch.bfloat16, device=q.device)      # K^T is for the backward
        vt = torch.empty((N, E, Lk), dtype=torch.bfloat16, device=q.device)
        with _lib.device_guard(q.device):
            _lib.check(lib.mpf_attn_transpose2(kb.data_ptr(), vb.data_ptr(), kt.data_ptr(), vt.data_ptr(), Lk, Lk, N, E, stream),
                       "mpf_attn_transpose2")
        m = None
        if mask is not None:
            m = mask.contiguous()
            assert m.dtype == torch.bool and m.shape[-2:] == (Lq, Lk)
        out = torch.empty((Lq, N, E), dtype=torch.bfloat16, device=q.device)
        lse = torch.empty((N, nheads, Lq), dtype=torch.float32, device=q.device)
        ws = _workspace(q.device, lib.mpf_attn_workspace_bytes(Lq, Lk, N, nheads))
        with _lib.device_guard(q.device):
            code = lib.mpf_attn_forward(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), m.data_ptr() if m is not None else None,
                                        1 if (m is not None and m.dim() == 3) else 0, out.data_ptr(), lse.data_ptr(),
                                        Lq, Lk, N, nheads, hd, 1.0 / math.sqrt(hd), ws.data_ptr(), ws.numel(), stream)
        _lib.check(code, "mpf_attn_forward")
        ctx.save_for_backward(qb, kb, vb, kt, m, out, lse)
        ctx.nheads = nheads
        ctx.in_dtypes = (q.dtype, k.dtype, v.dtype)
        return out.to(q.dtype)

    @staticmethod
    def backward(ctx, go):
        qb, kb, vb, kT, m, out, lse = ctx.saved_tensors
        H = ctx.nheads
        Lq, N, E = qb.shape
        Lk = kb.shape[0]
        hd = E // H
        dev = qb.device
        gob = go.to(torch.bfloat16).contiguous()
        lib = _lib.lib()
        stream = _lib.stream_ptr(dev)
        LqP = (Lq + 31) // 32 * 32
        delta = torch.empty((N, H, Lq), dtype=torch.float32, device=dev)
        qT = torch.empty((N, E, LqP), dtype=torch.bfloat16, device=dev)
        doT = torch.empty((N, E, LqP), dtype=torch.bfloat16, device=dev)
        dq = torch.empty_like(qb)
        dk = torch.empty_like(kb)
        dv = torch.empty_like(vb)
        ws = _workspace(dev, lib.mpf_attn_workspace_bytes(Lq, Lk, N, H))
        with _lib.device_guard(dev):
            _lib.check(lib.mpf_attn_bwd_prep(qb.data_ptr(), gob.data_ptr(), out.data_ptr(), qT.data_ptr(), doT.data_ptr(),
                                             delta.data_ptr(), Lq, LqP, N, H, stream), "mpf_attn_bwd_prep")      # Q^T, dO^T, delta
            code = lib.mpf_attn_backward(qb.data_ptr(), kb.data_ptr(), vb.data_ptr(), kT.data_ptr(), qT.data_ptr(),
                                         gob.data_ptr(), doT.data_ptr(), m.data_ptr() if m is not None else None,
                                         1 if (m is not None and m.dim() == 3) else 0, lse.data_ptr(), delta.data_ptr(),
                                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), Lq, LqP, Lk, N, H, hd,
                                         1.0 / math.sqrt(hd), ws.data_ptr(), ws.numel(), stream)
        _lib.check(code, "mpf_attn_backward")
        tq, tk, tv = ctx.in_dtypes
        return dq.to(tq), dk.to(tk), dv.to(tv), None, None


def _core_fp32(q, k, v, mask, nheads):
    """fp32 operands (no AMP): the same op in full precision with torch GEMMs — the native kernel is a
    bf16-operand MFMA kernel, and rounding fp32 inputs to bf16 would change the fp32 model's results."""
    Lq, N, E = q.shape
    Lk = k.shape[0]
    hd = E // nheads
    qh = q.reshape(Lq, N, nheads, hd).permute(1, 2, 0, 3)
    kh = k.reshape(Lk, N, nheads, hd).permute(1, 2, 0, 3)
    vh = v.reshape(Lk, N, nheads, hd).permute(1, 2, 0, 3)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if mask is not None:
        s = s.masked_fill(mask[:, None] if mask.dim() == 3 else mask[None, None], float("-inf"))
    return torch.matmul(torch.softmax(s, -1), vh).permute(2, 0, 1, 3).reshape(Lq, N, E)


def attention_core(q, k, v, mask, nheads):
    """bf16 / fp16 operands (the AMP training path) -> native MFMA kernels; fp32 operands -> fp32 math."""
    if not q.is_cuda:
        raise RuntimeError("mp_former_amd attention runs on the GPU only (no CPU fallback)")
    if q.dtype == torch.float32:
        return _core_fp32(q, k, v, mask, nheads)
    return _AttnCore.apply(q, k, v, mask, nheads)
