"""The mask predictions as their FACTORS (csrc/mask_fused.hip behind the C ABI).

    outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)      (mask2former_transformer_decoder.py:1865-1870)

is consumed on the training path only through point samples (matcher.py:120-132, criterion.py:141-191), so the decoder hands
the criterion a ``FactoredMasks`` (mask_embed rows + channel-last mask_features) instead of ten [N, Qtot, H/4, W/4] maps:
  * ``match_cost_fused``: the matcher's mask + dice cost straight from the factors (gather + interpolate the features at the
    points, MFMA product with the embeddings, cost sums in registers);
  * ``pair_planes``: the logits planes of the matched / mask-piloted (prediction, target) pairs only, as one MFMA launch over
    embedding rows gathered by the device assignment's indices, differentiable w.r.t. both factors (native backward);
  * ``full_product``: the same kernel over all rows — what a caller of the reference interface gets as ``pred_masks``.
bf16, 256 channels, GPU only (no CPU or eager fallback: unsupported shapes are rejected by ``supported`` before use).
"""
import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._h2d import upload


def _stream(dev):
    return _lib.stream_ptr(dev)


def _is_planes(x):
    """[N, C, H, W] whose images are dense [H*W, C] planes (channels_last), 16-byte aligned."""
    return (x.dim() == 4 and x.stride(1) == 1 and x.stride(3) == x.shape[1] and x.stride(2) == x.shape[3] * x.shape[1]
            and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def supported(me, mf):
    """me [N, R, 256] bf16 with contiguous rows, mf [N, 256, H, W] bf16 channel-last planes, H * W a multiple of 128."""
    return (me.is_cuda and mf.is_cuda and me.dtype == torch.bfloat16 and mf.dtype == torch.bfloat16 and me.dim() == 3
            and mf.dim() == 4 and me.shape[2] == 256 and mf.shape[1] == 256 and me.shape[0] == mf.shape[0] and me.stride(2) == 1
            and me.stride(0) % 8 == 0 and me.stride(1) % 8 == 0 and me.data_ptr() % 16 == 0 and _is_planes(mf)
            and (mf.shape[2] * mf.shape[3]) % 128 == 0)


class PairPlanes(Function):
    """planes[slot] = me_row(slot) . mf[image(slot)]  ->  [total_slots, H*W] bf16.
    ``row_off`` int64 (device): element offset from ``me.data_ptr()`` of the embedding row of every PAIR;
    ``pair_of_slot`` int32 (device, or None for the identity): pair of every slot; the slots of image b are
    ``slot_first[b] .. + slot_count[b]`` (int32, device).  Backward: d me (the paired rows; zero elsewhere) and d mf
    (channel-last planes), both native (mpf_pair_planes_backward)."""

    @staticmethod
    def forward(ctx, me, mf, row_off, pair_of_slot, slot_first, slot_count, total_slots, max_count):
        N, C, H, W = mf.shape
        HW = H * W
        dev = me.device
        out = torch.empty((total_slots, HW), dtype=torch.bfloat16, device=dev)
        if total_slots and max_count:
            with _lib.device_guard(dev):
                code = _lib.lib().mpf_pair_planes_forward(me.data_ptr(), row_off.data_ptr(),
                                                          pair_of_slot.data_ptr() if pair_of_slot is not None else None,
                                                          slot_first.data_ptr(), slot_count.data_ptr(), mf.data_ptr(), mf.stride(0),
                                                          out.data_ptr(), N, HW, C, _stream(dev))
            _lib.check(code, "mpf_pair_planes_forward")
        ctx.save_for_backward(me, mf, row_off, pair_of_slot if pair_of_slot is not None else row_off, slot_first, slot_count)
        ctx.identity = pair_of_slot is None
        ctx.sizes = (total_slots, max_count)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        me, mf, row_off, pair_of_slot, slot_first, slot_count = ctx.saved_tensors
        if ctx.identity:
            pair_of_slot = None
        total_slots, max_count = ctx.sizes
        N, C, H, W = mf.shape
        HW = H * W
        dev = me.device
        want_me, want_mf = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_me = torch.empty_strided(me.shape, me.stride(), dtype=me.dtype, device=dev).zero_() if want_me else None
        d_mf = torch.empty((N, HW, C), dtype=mf.dtype, device=dev) if want_mf else None
        if not (total_slots and max_count):
            if d_mf is not None:
                d_mf.zero_()
        else:
            g = g.contiguous()
            lib = _lib.lib()
            ws = torch.empty(lib.mpf_pair_planes_backward_workspace_bytes(N, HW, total_slots, max_count), dtype=torch.uint8, device=dev)
            with _lib.device_guard(dev):
                code = lib.mpf_pair_planes_backward(
                    g.data_ptr(), me.data_ptr(), row_off.data_ptr(), pair_of_slot.data_ptr() if pair_of_slot is not None else None,
                    slot_first.data_ptr(), slot_count.data_ptr(), mf.data_ptr(), mf.stride(0),
                    d_mf.data_ptr() if d_mf is not None else None, HW * C, d_me.data_ptr() if d_me is not None else None,
                    _lib.MPF_BF16, N, HW, C, total_slots, max_count, ws.data_ptr(), ws.numel(), _stream(dev))
            _lib.check(code, "mpf_pair_planes_backward")
        g_mf = d_mf.view(N, H, W, C).permute(0, 3, 1, 2) if d_mf is not None else None
        return d_me, g_mf, None, None, None, None, None, None


_identity_cache = {}


def _identity_layout(me, dev):
    """row offsets / slot ranges of the full product: slot b * R + r = row r of image b"""
    N, R, _ = me.shape
    key = (N, R, me.stride(0), me.stride(1), dev)
    lay = _identity_cache.get(key)
    if lay is None:
        b, r = np.meshgrid(np.arange(N, dtype=np.int64), np.arange(R, dtype=np.int64), indexing="ij")
        row_off = upload((b * me.stride(0) + r * me.stride(1)).reshape(-1), dev)
        i32 = upload(np.concatenate([np.arange(N) * R, np.full(N, R)]).astype(np.int32), dev)
        lay = (row_off, i32[:N], i32[N:])
        if len(_identity_cache) > 64:
            _identity_cache.clear()
        _identity_cache[key] = lay
    return lay


def full_product(me, mf):
    """einsum("bqc,bchw->bqhw", me, mf) -> [N, R, H, W] bf16 (differentiable), for ``supported`` operands."""
    N, R, _ = me.shape
    H, W = mf.shape[2:]
    row_off, first, count = _identity_layout(me, me.device)
    return PairPlanes.apply(me, mf, row_off, None, first, count, N * R, R).view(N, R, H, W)


class FactoredMasks:
    """Rows [q0, q1) of every image of ``einsum("bqc,bchw->bqhw", me, mf)``, unevaluated.  Behaves like the [N, q1 - q0, H, W]
    tensor where the decoder / criterion touch it (shape, dim-1 slicing, detach); anything else goes through
    ``materialize()`` (also reached by float() / cpu() / __torch_function__-free helpers used in the tests)."""

    def __init__(self, me, mf, q0=0, q1=None):
        self.me, self.mf = me, mf
        self.q0 = int(q0)
        self.q1 = int(me.shape[1] if q1 is None else q1)

    @property
    def shape(self):
        return torch.Size((self.me.shape[0], self.q1 - self.q0, self.mf.shape[2], self.mf.shape[3]))

    dtype = property(lambda self: self.me.dtype)
    device = property(lambda self: self.me.device)
    is_cuda = property(lambda self: self.me.is_cuda)
    requires_grad = property(lambda self: self.me.requires_grad or self.mf.requires_grad)

    def dim(self):
        return 4

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def same_factors(self, other):
        return isinstance(other, FactoredMasks) and other.me is self.me and other.mf is self.mf

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None) and isinstance(idx[1], slice):
            a, b, step = idx[1].indices(self.q1 - self.q0)
            if step == 1:
                return FactoredMasks(self.me, self.mf, self.q0 + a, self.q0 + max(a, b))
        return self.materialize()[idx]

    def detach(self):
        return FactoredMasks(self.me.detach(), self.mf.detach(), self.q0, self.q1)

    def materialize(self):
        return full_product(self.me[:, self.q0:self.q1], self.mf)

    def float(self):
        return self.materialize().float()

    def cpu(self):
        return self.materialize().cpu()

    def row_offsets(self, b, q):
        """numpy int64: element offsets (from me.data_ptr()) of the embedding rows of (image b, row q of this view)."""
        return b * self.me.stride(0) + (self.q0 + q) * self.me.stride(1)


def match_cost_fused(views, coords, tsamp, t_first, t_count, group_view, group_image, Q, Tmax, w_mask, w_dice):
    """Mask + dice matching cost (matcher.py:105-147) of G groups from the factors -> fp32 [G, Q, Tmax].
    views: FactoredMasks of one (me, mf); group g = rows of views[group_view[g]] in image group_image[g] (numpy int arrays);
    coords [G, P, 2]; tsamp [rows, P]; t_first / t_count numpy int [G]."""
    root = views[0]
    me, mf = root.me, root.mf
    dev = me.device
    G, P = coords.shape[0], coords.shape[1]
    H, W = mf.shape[2:]
    first = np.array([views[v].row_offsets(int(b), 0) for v, b in zip(group_view, group_image)], dtype=np.int64)
    first_d = upload(first, dev)
    i32 = upload(np.concatenate([group_image, t_first, t_count]).astype(np.int32), dev)
    cost = torch.zeros((G, Q, Tmax), dtype=torch.float32, device=dev)
    lib = _lib.lib()
    ws = torch.empty(lib.mpf_match_cost_fused_workspace_bytes(G, Q, Tmax, P, tsamp.shape[0]), dtype=torch.uint8, device=dev)
    with _lib.device_guard(dev):
        code = lib.mpf_match_cost_fused(me.data_ptr(), first_d.data_ptr(), me.stride(1), mf.data_ptr(), mf.stride(0), i32[:G].data_ptr(),
                                        H, W, mf.shape[1], coords.data_ptr(), tsamp.data_ptr(), tsamp.shape[0], i32[G:2 * G].data_ptr(),
                                        i32[2 * G:].data_ptr(), cost.data_ptr(), G, Q, Tmax, P, float(w_mask), float(w_dice),
                                        ws.data_ptr(), ws.numel(), _stream(dev))
    _lib.check(code, "mpf_match_cost_fused")
    return cost
