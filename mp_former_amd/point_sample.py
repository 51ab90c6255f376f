"""Native point sampling / fused point-sampled mask loss (csrc/loss.hip behind the C ABI).

Semantics (third-party, restated): detectron2.projects.point_rend.point_features
  point_sample(x, c) = grid_sample(x, 2c-1, bilinear, zeros, align_corners=False)
  get_uncertain_point_coords_with_randomness: draw oversample_ratio*P uniform points, keep the
  importance_sample_ratio*P most uncertain (uncertainty = -|logit|, criterion.py:73-87), append the
  remaining fresh uniform points
as used at mask2former/modeling/criterion.py:164-182 and matcher.py:122-132.

Maps of SEVERAL tensors (the prediction maps of all decoder layers) are addressed in one launch as
int64 element offsets from a common base pointer (`MapSet`).  CUDA tensors only; no CPU path.
"""
import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

_DT = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16, torch.uint8: _lib.MPF_U8, torch.bool: _lib.MPF_U8,
       "bits": _lib.MPF_BITS}
_CHUNKS = 8


def _stream(dev):
    return _lib.stream_ptr(dev)


def _row_slice_of(t):
    """If ``t`` [N, Q, h, w] is a row-range view (possibly through several views) of a dense parent that
    can be seen as [N, Qb, h, w] — e.g. the MP / matching halves of one layer's mask predictions inside
    the batched-heads tensor — return (root, Qb, first row); else None."""
    root = t._base
    if root is None or t.dim() != 4 or not root.is_contiguous():
        return None
    if t.requires_grad and not root.requires_grad:
        # a base outside the graph (e.g. a view made inside a custom Function's forward): addressing the maps through
        # it would silently drop their gradient
        return None
    N, Q, h, w = t.shape
    s0, s1, s2, s3 = t.stride()
    if s3 != 1 or s2 != w or s1 != h * w or s0 % (h * w) != 0:
        return None
    Qb = s0 // (h * w)
    d = t.storage_offset() - root.storage_offset()
    if d < 0 or root.numel() != N * Qb * h * w or d % (h * w) != 0:
        return None
    q0 = d // (h * w)
    if q0 + Q > Qb:
        return None
    return root, Qb, q0


class MapSet:
    """A list of [N, Q_i, h, w] tensors (dense [h,w] planes, arbitrary strides on dims 0/1) seen as one
    address space: offset(i, b, q) = element offset of that plane from the lowest data pointer.

    Tensors that are row-range views of a common parent (dim-1 slices) are addressed through the
    parent (``bases``): the gradient then goes to the parent in one piece instead of through one
    zero-padded slice-backward per view."""

    def __init__(self, tensors):
        t0 = tensors[0]
        if not t0.is_cuda:
            raise RuntimeError("mp_former_amd point sampling is implemented on the GPU only (no CPU fallback)")
        self.tensors = list(tensors)
        self.h, self.w = t0.shape[-2:]
        self.dtype, self.device = t0.dtype, t0.device
        self.esize = t0.element_size()
        if self.dtype not in _DT:
            raise RuntimeError(f"unsupported map dtype {self.dtype}")
        self.bases, self.base_of, self.q0 = [], [], []
        seen = {}
        for t in tensors:
            if t.dtype != self.dtype or tuple(t.shape[-2:]) != (self.h, self.w) or t.device != self.device:
                raise RuntimeError("all maps of a MapSet must share dtype, device and [h, w]")
            if t.stride(3) != 1 or t.stride(2) != self.w:
                raise RuntimeError("maps must be dense [h, w] planes")
            rs = _row_slice_of(t)
            if rs is None:
                k, q0 = id(t), 0
                if k not in seen:
                    seen[k] = len(self.bases)
                    self.bases.append(t)
            else:
                root, Qb, q0 = rs
                k = (id(root), Qb)
                if k not in seen:
                    seen[k] = len(self.bases)
                    self.bases.append(root.view(t.shape[0], Qb, self.h, self.w))     # one parent view per root
            self.base_of.append(seen[k])
            self.q0.append(q0)
        self.base_of = np.array(self.base_of, dtype=np.int64)
        self.q0 = np.array(self.q0, dtype=np.int64)
        ptrs = [b.data_ptr() for b in self.bases]
        self.base_ptr = min(ptrs)
        self.t_off = np.array([(p - self.base_ptr) // self.esize for p in ptrs], dtype=np.int64)
        self.s0 = np.array([b.stride(0) for b in self.bases], dtype=np.int64)
        self.s1 = np.array([b.stride(1) for b in self.bases], dtype=np.int64)
        # layout of one flat gradient buffer holding a CONTIGUOUS copy of every base tensor
        numels = np.array([b.numel() for b in self.bases], dtype=np.int64)
        self.g_start = np.concatenate([[0], np.cumsum(numels)[:-1]])
        self.g_total = int(numels.sum())
        self.g_s0 = np.array([b.shape[1] * self.h * self.w for b in self.bases], dtype=np.int64)

    def offsets(self, ti, b, q):
        """numpy int64 arrays (tensor index, image, query row) -> element offsets of the planes."""
        bi = self.base_of[ti]
        return self.t_off[bi] + b * self.s0[bi] + (q + self.q0[ti]) * self.s1[bi]

    def grad_offsets(self, ti, b, q):
        """element offsets of the same planes inside the flat gradient buffer of the bases"""
        bi = self.base_of[ti]
        return self.g_start[bi] + b * self.g_s0[bi] + (q + self.q0[ti]) * (self.h * self.w)


def point_sample_offsets(base_ptr, dtype, h, w, offs, coords, coord_rows, device):
    """out[i] = samples of the [h,w] plane at base_ptr + offs[i] (elements) at coords[coord_rows[i]]
    -> [n, P] f32.  No autograd."""
    n, P = offs.numel(), coords.shape[-2]
    out = torch.empty((n, P), dtype=torch.float32, device=device)
    if n == 0:
        return out
    with _lib.device_guard(device):
        code = _lib.lib().mpf_point_sample(base_ptr, _DT[dtype], h, w, offs.data_ptr(), coords.data_ptr(),
                                           coord_rows.data_ptr() if coord_rows is not None else None,
                                           out.data_ptr(), n, P, _stream(device))
    _lib.check(code, "mpf_point_sample")
    return out


def select_uncertain(vals, coords_in, k, P_out):
    """coords_out[:, :k] = coordinates of the k smallest-|value| entries per row (criterion.py:164-170)."""
    n, M = vals.shape
    out = torch.empty((n, P_out, 2), dtype=torch.float32, device=vals.device)
    if n and k:
        with _lib.device_guard(vals.device):
            code = _lib.lib().mpf_select_uncertain(vals.data_ptr(), coords_in.data_ptr(), out.data_ptr(), n, M, k, P_out,
                                                   _stream(vals.device))
        _lib.check(code, "mpf_select_uncertain")
    return out


def sample_select_uncertain(ms, offs, coords_in, k, P_out):
    """coords_out[:, :k] = the k most uncertain (smallest |logit|) of the candidate points of every row's
    plane.  bf16 planes of at most 128 KiB take the fused LDS kernel; everything else point_sample +
    select_uncertain."""
    n, M = coords_in.shape[0], coords_in.shape[1]
    if ms.dtype == torch.bfloat16 and ms.h * ms.w * 2 <= 128 * 1024 and (ms.h * ms.w * 2) % 16 == 0 and M <= 40960 and n and k:
        out = torch.empty((n, P_out, 2), dtype=torch.float32, device=coords_in.device)
        with _lib.device_guard(ms.device):
            code = _lib.lib().mpf_sample_select_uncertain(ms.base_ptr, _DT[ms.dtype], ms.h, ms.w, offs.data_ptr(), coords_in.data_ptr(),
                                                          out.data_ptr(), n, M, k, P_out, _stream(ms.device))
        _lib.check(code, "mpf_sample_select_uncertain")
        return out
    logits = point_sample_offsets(ms.base_ptr, ms.dtype, ms.h, ms.w, offs, coords_in, None, ms.device)
    return select_uncertain(logits, coords_in, k, P_out)


def match_cost(ms, offs, coords, coord_rows, tsamp, t_first, t_count, Tmax, w_mask, w_dice, rows_per_group=1):
    n = offs.numel()
    P = coords.shape[-2]
    cost = torch.zeros((n, Tmax), dtype=torch.float32, device=ms.device)
    if n:
        with _lib.device_guard(ms.device):
            code = _lib.lib().mpf_match_cost(ms.base_ptr, _DT[ms.dtype], ms.h, ms.w, offs.data_ptr(), coords.data_ptr(),
                                             coord_rows.data_ptr(), tsamp.data_ptr(), t_first.data_ptr(),
                                             t_count.data_ptr(), cost.data_ptr(), n, Tmax, P, float(w_mask),
                                             float(w_dice), int(rows_per_group), _stream(ms.device))
        _lib.check(code, "mpf_match_cost")
    return cost


def _mask_loss_sums_forward(ctx, ms, pred_offs, gt, gt_rows, coords):
    """forward of MaskLossSums / MaskLossSumsCompact; leaves (gt_u8, gt_hw, gdt, ms) on ctx"""
    # gt: GTMasks (its bit-packed copy is used when present) or a [R, H, W] byte tensor
    gt_u8 = gt if torch.is_tensor(gt) else (gt.bits if gt.bits is not None else gt.u8)
    gdt = _lib.MPF_U8 if (torch.is_tensor(gt) or gt.bits is None) else _lib.MPF_BITS
    n, P = coords.shape[0], coords.shape[1]
    H, W = (gt.shape[-2:] if torch.is_tensor(gt) else (gt.H, gt.W))
    ctx.gt_hw, ctx.gdt, ctx.gt_u8 = (H, W), gdt, gt_u8
    partial = torch.empty((n, _CHUNKS, 4), dtype=torch.float32, device=ms.device)
    if n:
        with _lib.device_guard(ms.device):
            code = _lib.lib().mpf_mask_loss_forward(
                ms.base_ptr, _DT[ms.dtype], ms.h, ms.w, pred_offs.data_ptr(), gt_u8.data_ptr(), gdt, H, W,
                gt_rows.data_ptr(), coords.data_ptr(), partial.data_ptr(), n, P, _CHUNKS, _stream(ms.device))
        _lib.check(code, "mpf_mask_loss_forward")
    ctx.ms = ms
    return partial.sum(1)


class MaskLossSums(Function):
    """sums[i] = (sum_p BCE(x,t), sum_p sigmoid(x)*t, sum_p sigmoid(x), sum_p t) over the P points of
    pair i; x sampled from the plane at ms.base + pred_offs[i], t from gt_u8[gt_rows[i]]
    (criterion.py:172-191).  Differentiable wrt every base tensor of the MapSet (pass ``*ms.bases``).

    Backward: one workgroup per (pair, band of its plane) accumulates the bilinear scatter in LDS and
    writes the band once into a dense zero-initialised gradient of the maps' dtype (the planes of a
    step's pairs are distinct: a query is matched once per output) — no global atomics, no fp32 image
    of every map."""

    @staticmethod
    def forward(ctx, ms, pred_offs, grad_offs, gt, gt_rows, coords, *tensors):
        sums = _mask_loss_sums_forward(ctx, ms, pred_offs, gt, gt_rows, coords)
        ctx.save_for_backward(pred_offs, grad_offs, ctx.gt_u8, gt_rows, coords, *tensors)   # keeps the maps alive
        return sums

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sums):
        ms = ctx.ms
        pred_offs, grad_offs, gt_u8, gt_rows, coords = ctx.saved_tensors[:5]
        n, P = coords.shape[0], coords.shape[1]
        H, W = ctx.gt_hw
        gbuf = torch.zeros(ms.g_total, dtype=ms.dtype, device=ms.device)
        if n:
            g = grad_sums.contiguous().float()
            with _lib.device_guard(ms.device):
                code = _lib.lib().mpf_mask_loss_backward_dense(
                    ms.base_ptr, _DT[ms.dtype], ms.h, ms.w, pred_offs.data_ptr(), gt_u8.data_ptr(), ctx.gdt, H, W,
                    gt_rows.data_ptr(), coords.data_ptr(), g.data_ptr(), gbuf.data_ptr(), _DT[ms.dtype], grad_offs.data_ptr(),
                    n, P, _stream(ms.device))
            _lib.check(code, "mpf_mask_loss_backward_dense")
        grads = []
        for i, t in enumerate(ms.bases):
            s = int(ms.g_start[i])
            grads.append(gbuf[s:s + t.numel()].view(t.shape))
        return (None, None, None, None, None, None, *grads)


class MaskLossSumsPlanes(Function):
    """MaskLossSums over the compact planes of the step's pairs (mask_fused.pair_planes: [slots, h*w], one plane per
    pair): the backward scatter writes every paired plane of a fresh gradient buffer exactly once (one workgroup per
    (pair, band): LDS fixed-point accumulators, no atomics on memory), so there is no zero-fill; rows of padding slots
    stay uninitialised and are never read (the products of pair_planes' backward run on the paired rows only)."""

    @staticmethod
    def forward(ctx, ms, plane_offs, gt, gt_rows, coords, planes):
        sums = _mask_loss_sums_forward(ctx, ms, plane_offs, gt, gt_rows, coords)
        ctx.save_for_backward(plane_offs, ctx.gt_u8, gt_rows, coords, planes)
        return sums

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sums):
        ms = ctx.ms
        plane_offs, gt_u8, gt_rows, coords, planes = ctx.saved_tensors
        n, P = coords.shape[0], coords.shape[1]
        H, W = ctx.gt_hw
        g_planes = torch.empty_like(planes)
        if n:
            g = grad_sums.contiguous().float()
            with _lib.device_guard(ms.device):
                code = _lib.lib().mpf_mask_loss_backward_dense(
                    ms.base_ptr, _DT[ms.dtype], ms.h, ms.w, plane_offs.data_ptr(), gt_u8.data_ptr(), ctx.gdt, H, W,
                    gt_rows.data_ptr(), coords.data_ptr(), g.data_ptr(), g_planes.data_ptr(), _DT[ms.dtype], plane_offs.data_ptr(),
                    n, P, _stream(ms.device))
            _lib.check(code, "mpf_mask_loss_backward_dense")
        return None, None, None, None, None, g_planes
