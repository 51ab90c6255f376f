"""Native point sampling / fused point-sampled mask loss (csrc/loss.hip behind the C ABI).

Semantics (third-party, restated): detectron2.projects.point_rend.point_features
  point_sample(x, c) = grid_sample(x, 2c-1, bilinear, zeros, align_corners=False)
  get_uncertain_point_coords_with_randomness: draw oversample_ratio*P uniform points, keep the
  importance_sample_ratio*P most uncertain (uncertainty = -|logit|, criterion.py:73-87), append the
  remaining fresh uniform points
as used at mask2former/modeling/criterion.py:164-182 and matcher.py:122-132.

All functions need CUDA tensors; there is no CPU path.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, _rng

_DT = {torch.float32: _lib.MPF_F32, torch.bfloat16: _lib.MPF_BF16, torch.uint8: _lib.MPF_U8, torch.bool: _lib.MPF_U8}
_CHUNKS = 8


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mp_former_amd point sampling is implemented on the GPU only (no CPU fallback)")


def map_rows(maps, index):
    """Row indices (in units of one h*w map, relative to maps.data_ptr()) of maps[index] for a 4-d
    [N,Q,h,w] tensor that may be a slice of a larger contiguous tensor along dim 1.
    index = (b[n], q[n]) int64 tensors on the same device."""
    N, Q, h, w = maps.shape
    s0, s1, s2, s3 = maps.stride()
    if s3 != 1 or s2 != w or s0 % (h * w) or s1 % (h * w):
        raise RuntimeError("prediction maps must be dense [*, *, h, w] planes")
    b, q = index
    return (b * (s0 // (h * w)) + q * (s1 // (h * w))).to(torch.int32)


def point_sample_rows(src, h, w, rows, coords, coord_rows=None):
    """out[i] = bilinear samples of map rows[i] of `src` (any tensor whose memory is [*, h, w] planes
    starting at src.data_ptr()) at coords[coord_rows[i]] -> [n, P] f32.  No autograd."""
    _need_cuda(src, rows, coords, coord_rows)
    if src.dtype not in _DT:
        raise RuntimeError(f"point_sample: unsupported dtype {src.dtype}")
    coords = coords.contiguous().float()
    n, P = rows.numel(), coords.shape[-2]
    out = torch.empty((n, P), dtype=torch.float32, device=src.device)
    if n == 0:
        return out
    with torch.cuda.device(src.device):
        code = _lib.lib().mpf_point_sample(src.data_ptr(), _DT[src.dtype], h, w, rows.data_ptr(), coords.data_ptr(),
                                           coord_rows.data_ptr() if coord_rows is not None else None,
                                           out.data_ptr(), n, P, _stream(src))
    _lib.check(code, "mpf_point_sample")
    return out


class MaskLossSums(Function):
    """sums[i] = (sum_p BCE(x,t), sum_p sigmoid(x)*t, sum_p sigmoid(x), sum_p t) over the P points of
    pair i, x / t sampled from pred[pred_rows[i]] / gt[gt_rows[i]]  (criterion.py:172-191)."""

    @staticmethod
    def forward(ctx, pred, pred_rows, gt_u8, gt_rows, coords):
        _need_cuda(pred, pred_rows, gt_u8, gt_rows, coords)
        h, w = pred.shape[-2:]
        H, W = gt_u8.shape[-2:]
        n, P = coords.shape[0], coords.shape[1]
        partial = torch.empty((n, _CHUNKS, 4), dtype=torch.float32, device=pred.device)
        if n:
            with torch.cuda.device(pred.device):
                code = _lib.lib().mpf_mask_loss_forward(
                    pred.data_ptr(), _DT[pred.dtype], h, w, pred_rows.data_ptr(), gt_u8.data_ptr(), H, W,
                    gt_rows.data_ptr(), coords.data_ptr(), partial.data_ptr(), n, P, _CHUNKS, _stream(pred))
            _lib.check(code, "mpf_mask_loss_forward")
        ctx.save_for_backward(pred, pred_rows, gt_u8, gt_rows, coords)
        return partial.sum(1)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sums):
        pred, pred_rows, gt_u8, gt_rows, coords = ctx.saved_tensors
        h, w = pred.shape[-2:]
        H, W = gt_u8.shape[-2:]
        n, P = coords.shape[0], coords.shape[1]
        # fp32 accumulation buffer with pred's memory layout (pred may be a strided slice)
        gbuf = torch.zeros(pred.shape, dtype=torch.float32, device=pred.device) if pred.is_contiguous() else \
            torch.empty_strided(pred.shape, pred.stride(), dtype=torch.float32, device=pred.device).zero_()
        if n:
            g = grad_sums.contiguous().float()
            with torch.cuda.device(pred.device):
                code = _lib.lib().mpf_mask_loss_backward(
                    pred.data_ptr(), _DT[pred.dtype], h, w, pred_rows.data_ptr(), gt_u8.data_ptr(), H, W,
                    gt_rows.data_ptr(), coords.data_ptr(), g.data_ptr(), gbuf.data_ptr(), n, P, _stream(pred))
            _lib.check(code, "mpf_mask_loss_backward")
        return gbuf.to(pred.dtype), None, None, None, None


def uncertain_point_coords(pred, pred_rows, num_points, oversample_ratio, importance_sample_ratio, tag):
    """criterion.py:164-170 -> coords [n, num_points, 2] (no grad)."""
    assert oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1
    n = pred_rows.numel()
    dev = pred.device
    h, w = pred.shape[-2:]
    num_sampled = int(num_points * oversample_ratio)
    coords = _rng.rand(tag + "_over", (n, num_sampled, 2), dev)
    logits = point_sample_rows(pred, h, w, pred_rows, coords)
    num_uncertain = int(importance_sample_ratio * num_points)
    num_random = num_points - num_uncertain
    idx = torch.topk(-logits.abs(), k=num_uncertain, dim=1)[1]
    picked = torch.gather(coords, 1, idx[:, :, None].expand(-1, -1, 2))
    if num_random > 0:
        picked = torch.cat([picked, _rng.rand(tag + "_rand", (n, num_random, 2), dev)], dim=1)
    return picked.contiguous()
