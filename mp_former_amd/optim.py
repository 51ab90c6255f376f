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
        self._beta1, self._beta2 = (float(x) for x in self.param_groups[0]["betas"])
        self._key = None
        self._scratch = None
        self.norm_clip = None          # device tensor [2] = (gradient norm, clip coefficient) of the last step

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # (one pass over the parameters: this loop runs ~350 times per step on the launch thread)
        plist, grads = [], []
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is not None and p.numel():
                    plist.append((p, group))
                    grads.append(g)
        if not plist:
            return loss
        key = tuple(map(id, [p for p, _ in plist]))
        if key != self._key:
            self._build(plist, key)
        keep, glist = [], []
        f32 = torch.float32
        for (p, _), g in zip(plist, grads):
            if g.dtype is not f32 or g.is_sparse or g.device != p.device:
                if g.is_sparse:
                    raise RuntimeError("ClipAdamW does not support sparse gradients")
                raise RuntimeError("mp_former_amd ClipAdamW: fp32 gradients on the parameter's device only")
            if g.stride() != p.stride():          # the update is element-wise over raw memory: same layout needed
                g = torch.empty_like(p).copy_(g)
                keep.append(g)
            glist.append(g.data_ptr())
        t = self._table
        t["grad"] = glist
        t["lr"] = [g["lr"] for g in self._groups]                 # schedulers rewrite these between steps
        t["wd"] = [g["weight_decay"] for g in self._groups]
        self._steps += 1.0
        t["bc1"] = 1.0 - self._beta1 ** self._steps
        t["bc2_sqrt"] = np.sqrt(1.0 - self._beta2 ** self._steps)
        dev = plist[0][0].device
        items = upload(t.view(np.int64).reshape(-1), dev)
        blk = self._blocks
        if self._scratch is None or self._scratch.numel() < blk or self._scratch.device != dev:
            self._scratch = torch.empty(blk, dtype=torch.float32, device=dev)
        self.norm_clip = torch.empty(2, dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            code = _lib.lib().mpf_clip_adamw_step(items.data_ptr(), len(plist), blk, self.max_norm, self._beta1, self._beta2,
                                                  self.param_groups[0]["eps"], self._scratch.data_ptr(),
                                                  self.norm_clip.data_ptr(), _lib.stream_ptr(dev))
        _lib.check(code, "mpf_clip_adamw_step")
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad without its per-call bookkeeping (profiler scope, foreach grouping): 0.4 -> 0.1 ms of
        launch-thread time for ~350 parameters"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def _build(self, plist, key):
        """static part of the launch table (parameter / moment pointers, sizes, lr, weight decay) for this set of
        parameters; rebuilt only when the set of parameters that have a gradient changes"""
        self._sync_steps()
        rows, blk, dev = [], 0, plist[0][0].device
        steps = []
        for p, group in plist:
            if not p.is_cuda or p.dtype != torch.float32 or not _dense(p):
                raise RuntimeError("mp_former_amd ClipAdamW: dense fp32 CUDA parameters only (no CPU fallback)")
            if p.device != dev:
                raise RuntimeError("ClipAdamW: all parameters must live on one device")
            st = self.state[p]
            if not st:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p)           # preserve_format: the parameter's layout
                st["exp_avg_sq"] = torch.zeros_like(p)
            steps.append(float(st["step"]))
            rows.append((p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk,
                         group["lr"], group["weight_decay"], 1.0, 1.0))
            blk += (p.numel() + 2047) // 2048
        self._table = np.array(rows, dtype=_ITEM)
        self._steps = np.asarray(steps, dtype=np.float64)
        self._plist = [p for p, _ in plist]
        self._groups = [g for _, g in plist]
        self._blocks = blk
        self._key = key

    def _sync_steps(self):
        """write the host-side step counts back into the per-parameter state (state_dict layout of torch.optim.AdamW)"""
        if self._key is not None:
            for p, n in zip(self._plist, self._steps):
                self.state[p]["step"] = torch.tensor(float(n), dtype=torch.float32)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._key = None          # moments were replaced: rebuild the table on the next step
