"""Hungarian matcher — mirror of mask2former/modeling/matcher.py (:15-62 cost functions, :70-179
HungarianMatcher).

The matchings of ALL decoder outputs of a step (final + auxiliary) are computed together:
  * one native sampling launch evaluates every ground-truth byte mask at every output's point set,
  * one native launch (mpf_match_cost) produces the mask + dice cost of every (output, image, query,
    target) from the prediction maps in place — no gathered / float copies, no [Q,P] temporaries,
  * the class cost is one batched softmax + gather,
  * the assignment itself: the training criterion solves all problems on the device (csrc/lsa.hip,
    SciPy's algorithm and tie-breaking, no device->host copy at all); `match_many` / `forward` (the
    reference interface, CPU index tensors) make ONE device->host copy for all outputs and call SciPy's
    linear_sum_assignment as the reference does (:149-151) — either way the 10*N blocking `.cpu()`
    calls per step are gone.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from . import _rng
from . import _lib
from ._h2d import upload
from ._targets import stacked_masks
from .mask_fused import FactoredMasks, match_cost_fused
from .point_sample import MapSet, match_cost, point_sample_offsets


class GTMasks:
    """All ground-truth masks of a batch as one byte tensor [sum T_b, H, W] (0/1), built once per
    step; `offsets[b]` is the first row of image b."""

    def __init__(self, targets):
        dev = targets[0]["masks"].device
        if not dev.type == "cuda":
            raise RuntimeError("mp_former_amd criterion / matcher run on the GPU only (no CPU fallback)")
        self.counts = [int(t["masks"].shape[0]) for t in targets]
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.total = self.offsets[-1]
        self.tmax = max(self.counts) if self.counts else 0
        ms = [t["masks"] for t in targets if t["masks"].shape[0] > 0]
        if ms:
            if all(x.dtype == torch.bool and x.shape[1:] == ms[0].shape[1:] for x in ms):
                m = stacked_masks(ms)             # the decoder's mask-piloted setup has usually stacked them already
            else:
                m = torch.cat([x if x.dtype == torch.bool else (x > 0) for x in ms]).contiguous()
            self.u8 = m.view(torch.uint8)
        else:
            H, W = targets[0]["masks"].shape[-2:]
            self.u8 = torch.zeros((0, H, W), dtype=torch.uint8, device=dev)
        self.H, self.W = self.u8.shape[-2:]
        # bit-packed copy (32 pixels per word) for the loss / matcher gathers: 8x smaller, stays in L2
        self.bits = None
        if self.total and (self.H * self.W) % 32 == 0:
            self.bits = torch.empty((self.total, self.H * self.W // 32), dtype=torch.int32, device=dev)
            with _lib.device_guard(dev):
                _lib.check(_lib.lib().mpf_pack_mask_bits(self.u8.data_ptr(), self.bits.data_ptr(), self.u8.numel(),
                                                         _lib.stream_ptr(dev)), "mpf_pack_mask_bits")
        self.image_of_row = np.concatenate([np.full(c, b, dtype=np.int64) for b, c in enumerate(self.counts)]) \
            if self.total else np.zeros(0, dtype=np.int64)
        self.device = dev


class HungarianMatcher(nn.Module):
    # True: the mask / dice costs are sampled from the MATERIALISED prediction maps in the dtype the mask product returns them
    # in — under bf16 autocast the bf16-ROUNDED maps, interpolated in fp32, which is exactly what the reference's matcher reads
    # (matcher.py:120-132 after the autocast einsum of mask2former_transformer_decoder.py:1865) — instead of from the factors
    # (fp32-class logits, csrc/mask_fused.hip).  Costs agree to ~2^-9 relative; an assignment that hinges on less can differ
    # (tests/test_head_gpu.py::test_amp_matching_on_bf16_rounded_maps_as_the_reference_samples_them measures how many do).
    # Slower (the [N, Q, H/4, W/4] maps of all outputs are written and sampled): for trajectory comparisons, not for speed.
    reference_amp_rounding = False

    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.num_points = num_points

    @torch.no_grad()
    def cost_matrices(self, outs, targets, gt=None, tags=None, mapset=None, map_index=None):
        """The matching cost of every (output, image, query, target) (matcher.py:95-147), on the device:
        fp32 [L, N, Q, Tmax] (columns >= the image's target count are padding), or None when no image
        has a target.  Draws the matcher's point sets either way (RNG parity)."""
        L = len(outs)
        N, Q = outs[0]["pred_logits"].shape[:2]
        gt = gt or GTMasks(targets)
        dev = gt.device
        P = self.num_points
        tags = tags or ["match"] + [f"match_{i}" for i in range(L - 1)]
        # one point set per (output, image), shared by all of that image's masks (matcher.py:120)
        coords = _rng.rand_cat([(tags[l], (1, P, 2)) for l in range(L) for _ in range(N)], dev)   # [L*N,P,2]
        if gt.tmax == 0:
            return None
        Tt, Tmax = gt.total, gt.tmax
        counts = np.asarray(gt.counts, dtype=np.int64)
        firsts = np.asarray(gt.offsets[:-1], dtype=np.int64)
        # ground-truth samples [L*Tt, P]: every mask at the point set of its (output, image)
        gt_offs = np.tile(np.arange(Tt, dtype=np.int64) * (gt.H * gt.W), L)
        gcrow = (np.repeat(np.arange(L), Tt) * N + np.tile(gt.image_of_row, L)).astype(np.int32)
        masks = [o["pred_masks"] for o in outs]
        factored = (mapset is None and not self.reference_amp_rounding and Tmax <= 128
                    and all(isinstance(m, FactoredMasks) and m.same_factors(masks[0]) for m in masks))
        if factored:
            # the maps are kept as (mask_embed, mask_features): cost straight from the factors (csrc/mask_fused.hip)
            gt_offs_d = upload(gt_offs, dev)
            gcrow_d = upload(gcrow, dev)
            tsamp = self._gt_samples(gt, gt_offs_d, coords, gcrow_d, dev)
            l_idx, b_idx = np.meshgrid(np.arange(L), np.arange(N), indexing="ij")
            l_idx, b_idx = l_idx.reshape(-1), b_idx.reshape(-1)
            C = match_cost_fused(masks, coords, tsamp, l_idx * Tt + firsts[b_idx], counts[b_idx], l_idx, b_idx, Q, Tmax,
                                 self.cost_mask, self.cost_dice).view(L, N, Q, Tmax)
        else:
            if mapset is None:
                mapset = MapSet([m.materialize() if isinstance(m, FactoredMasks) else m for m in masks])
                map_index = list(range(L))
            # ---- index arrays (host) -> one upload ---------------------------------------------------
            l_idx, b_idx, q_idx = np.meshgrid(np.arange(L), np.arange(N), np.arange(Q), indexing="ij")
            l_idx, b_idx, q_idx = l_idx.reshape(-1), b_idx.reshape(-1), q_idx.reshape(-1)
            pred_offs = mapset.offsets(np.asarray(map_index, dtype=np.int64)[l_idx], b_idx, q_idx)
            i64 = upload(np.concatenate([pred_offs, gt_offs]), dev)
            i32 = np.concatenate([
                (l_idx * N + b_idx),                                        # coord row of every prediction row
                (l_idx * Tt + firsts[b_idx]),                               # first tsamp row of its image
                counts[b_idx],                                              # number of targets of its image
                gcrow,                                                      # coord row of every GT sample row
            ]).astype(np.int32)
            i32 = upload(i32, dev)
            n_rows = L * N * Q
            pred_offs_d, gt_offs_d = i64[:n_rows], i64[n_rows:]
            crow_d, tfirst_d, tcount_d, gcrow_d = i32[:n_rows], i32[n_rows:2 * n_rows], i32[2 * n_rows:3 * n_rows], i32[3 * n_rows:]
            # ---- ground-truth samples [L*Tt, P], then mask + dice cost [L*N*Q, Tmax] ------------------
            tsamp = self._gt_samples(gt, gt_offs_d, coords, gcrow_d, dev)
            C = match_cost(mapset, pred_offs_d, coords, crow_d, tsamp, tfirst_d, tcount_d, Tmax,
                           self.cost_mask, self.cost_dice, rows_per_group=Q).view(L, N, Q, Tmax)   # rows are (l, b, q), q fastest
        # ---- class cost: -softmax(logits)[:, labels]  (matcher.py:105-111) ------------------------
        logits = torch.stack([o["pred_logits"] for o in outs]).float()                  # [L,N,Q,K+1]
        labels = torch.zeros((N, Tmax), dtype=torch.int64, device=dev)
        for b, t in enumerate(targets):
            if gt.counts[b]:
                labels[b, :gt.counts[b]] = t["labels"]
        prob = logits.softmax(-1)
        return (C - self.cost_class * torch.gather(prob, 3, labels[None, :, None, :].expand(L, N, Q, Tmax))).contiguous()

    @staticmethod
    def _gt_samples(gt, gt_offs_d, coords, gcrow_d, dev):
        if gt.bits is not None:      # gt_offs are pixel offsets either way
            return point_sample_offsets(gt.bits.data_ptr(), "bits", gt.H, gt.W, gt_offs_d, coords, gcrow_d, dev)
        return point_sample_offsets(gt.u8.data_ptr(), torch.uint8, gt.H, gt.W, gt_offs_d, coords, gcrow_d, dev)

    @torch.no_grad()
    def match_many(self, outs, targets, gt=None, tags=None, mapset=None, map_index=None):
        """outs: list of {"pred_logits" [N,Q,K+1], "pred_masks" [N,Q,h,w]}  ->  list (per output) of
        list (per image) of (index_i, index_j) int64 CPU tensors (matcher.py:95-156), solved by SciPy on
        the host as in the reference (one device->host copy for all outputs).  The training criterion
        uses the device solver instead (criterion.py, lsa.py) and never synchronises.
        `mapset` / `map_index` optionally give a MapSet that already contains the mask tensors."""
        L = len(outs)
        N = outs[0]["pred_logits"].shape[0]
        gt = gt or GTMasks(targets)
        empty = (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64))
        C = self.cost_matrices(outs, targets, gt=gt, tags=tags, mapset=mapset, map_index=map_index)
        if C is None:
            return [[empty for _ in range(N)] for _ in range(L)]
        host = C.cpu().numpy()                                                           # the one D2H copy
        res = []
        for l in range(L):
            cur = []
            for b in range(N):
                if gt.counts[b] == 0:
                    cur.append(empty)
                    continue
                i, j = linear_sum_assignment(host[l, b, :, :gt.counts[b]])
                cur.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            res.append(cur)
        return res

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Reference interface: one output dict -> list of (index_i, index_j) per image."""
        return self.match_many([{"pred_logits": outputs["pred_logits"], "pred_masks": outputs["pred_masks"]}], targets)[0]

    def __repr__(self, _repr_indent=4):
        body = [f"cost_class: {self.cost_class}", f"cost_mask: {self.cost_mask}", f"cost_dice: {self.cost_dice}"]
        return "\n".join(["Matcher " + self.__class__.__name__] + [" " * _repr_indent + line for line in body])
