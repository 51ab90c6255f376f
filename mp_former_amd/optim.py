"""Optimizer tail of the training step: AdamW with full-model gradient-norm
clipping — the reference's ``FullModelGradientClippingOptimizer`` around ``torch.optim.AdamW``
(train_net.py:259-337; :316-320: ``clip_grad_norm_(all parameters, CLIP_VALUE)`` then ``AdamW.step()``) —
as three native launches over all parameters (csrc/elementwise.hip, mpf_clip_adamw_step): squared-norm
partials, a fixed-order reduction that derives the clip coefficient on the device, and the AdamW update
applying it on the fly.  No host synchronisation, no extra pass that rewrites the gradients; same
arithmetic as ``torch.nn.utils.clip_grad_norm_`` + torch's fused AdamW (tests/test_optim_gpu.py).

A parameter whose ``.grad`` is None is skipped for that step and keeps its own step count (per-parameter
bias corrections, as torch).  ``state_dict`` has torch.optim.AdamW's layout (step / exp_avg / exp_avg_sq
per parameter; the step counts are host numbers here).  GPU only, fp32 parameters.
"""
import numpy as np
import torch

from . import _lib
from ._h2d import upload

_ITEM = np.dtype([("param", "<u8"), ("grad", "<u8"), ("m", "<u8"), ("v", "<u8"), ("numel", "<i8"), ("first", "<i8"),
                  ("lr", "<f4"), ("wd", "<f4"), ("bc1", "<f4"), ("bc2_sqrt", "<f4")])
assert _ITEM.itemsize == 64


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        b = {tuple(g["betas"]) for g in self.param_groups} | {g["eps"] for g in self.param_groups}
        if len(b) != 2:
            raise ValueError("ClipAdamW: betas and eps must be the same in every parameter group")
        self.max_norm = float(max_norm)
        self._scratch = None
        self.norm_clip = None          # device tensor [2] = (gradient norm, clip coefficient) of the last step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rows, keep = [], []
        blk = 0
        dev = None
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None or p.numel() == 0:
                    continue
                if g.is_sparse:
                    raise RuntimeError("ClipAdamW does not support sparse gradients")
                if not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or not _dense(p):
                    raise RuntimeError("mp_former_amd ClipAdamW: dense fp32 CUDA parameters only (no CPU fallback)")
                if g.stride() != p.stride():          # the update is element-wise over raw memory: same layout needed
                    g = torch.empty_like(p).copy_(g)
                    keep.append(g)
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p)           # preserve_format: the parameter's layout
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                t = float(st["step"])
                dev = p.device if dev is None else dev
                if p.device != dev:
                    raise RuntimeError("ClipAdamW: all parameters must live on one device")
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk,
                             group["lr"], group["weight_decay"], 1.0 - beta1 ** t, (1.0 - beta2 ** t) ** 0.5))
                blk += (p.numel() + 2047) // 2048
        if not rows:
            return loss
        table = np.array(rows, dtype=_ITEM)
        items = upload(table.view(np.int64).reshape(-1), dev)
        if self._scratch is None or self._scratch.numel() < blk or self._scratch.device != dev:
            self._scratch = torch.empty(blk, dtype=torch.float32, device=dev)
        self.norm_clip = torch.empty(2, dtype=torch.float32, device=dev)
        beta1, beta2 = self.param_groups[0]["betas"]
        with torch.cuda.device(dev):
            code = _lib.lib().mpf_clip_adamw_step(items.data_ptr(), len(rows), blk, self.max_norm, beta1, beta2,
                                                  self.param_groups[0]["eps"], self._scratch.data_ptr(),
                                                  self.norm_clip.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(code, "mpf_clip_adamw_step")
        return loss

