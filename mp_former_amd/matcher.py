"""Hungarian matcher — mirror of mask2former/modeling/matcher.py (:15-62 cost functions, :70-179
HungarianMatcher).  The point samples of predictions and byte ground-truth masks come from the native
sampling kernel (csrc/loss.hip); the [Q,T] cost matrices are small fp32 GEMMs; the assignment is
SciPy's linear_sum_assignment on the host exactly as in the reference (:149-151), but ALL cost
matrices of a step travel in ONE device->host copy instead of one blocking `.cpu()` per image per
decoder layer — assignments are unchanged.
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from . import _rng
from .point_sample import map_rows, point_sample_rows


class GTMasks:
    """All ground-truth masks of a batch as one byte tensor [sum T_b, H, W] (0/1), built once per
    step; `offsets[b]` is the first row of image b."""

    def __init__(self, targets):
        dev = targets[0]["masks"].device
        self.counts = [int(t["masks"].shape[0]) for t in targets]
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        ms = [t["masks"] for t in targets if t["masks"].shape[0] > 0]
        if ms:
            m = torch.cat([x if x.dtype == torch.bool else (x > 0) for x in ms]).contiguous()
            self.u8 = m.view(torch.uint8)
        else:
            H, W = targets[0]["masks"].shape[-2:]
            self.u8 = torch.zeros((0, H, W), dtype=torch.uint8, device=dev)
        self.H, self.W = self.u8.shape[-2:]
        total = self.offsets[-1]
        self.rows = torch.arange(total, dtype=torch.int32, device=dev)
        self.image_of_row = torch.cat([torch.full((c,), b, dtype=torch.int32) for b, c in enumerate(self.counts)]).to(dev) \
            if total else torch.zeros(0, dtype=torch.int32, device=dev)


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.num_points = num_points

    @torch.no_grad()
    def cost_matrices(self, outputs, targets, tag="match", gt=None):
        """[C_b of shape [Q, T_b]] on the device (matcher.py:103-148)."""
        logits, masks = outputs["pred_logits"], outputs["pred_masks"]
        bs, Q = logits.shape[:2]
        dev = masks.device
        h, w = masks.shape[-2:]
        gt = gt or GTMasks(targets)
        P = self.num_points
        # one point set per image, shared by all of its masks (matcher.py:120)
        coords = torch.cat([_rng.rand(tag, (1, P, 2), dev) for _ in range(bs)], 0)
        bq = torch.arange(bs * Q, device=dev)
        rows = map_rows(masks, (bq // Q, bq % Q))
        out_pts = point_sample_rows(masks, h, w, rows, coords, (bq // Q).to(torch.int32)).view(bs, Q, P)
        tgt_pts = point_sample_rows(gt.u8, gt.H, gt.W, gt.rows, coords, gt.image_of_row)
        prob = logits.float().softmax(-1)
        costs = []
        for b in range(bs):
            o = out_pts[b]
            t = tgt_pts[gt.offsets[b]:gt.offsets[b + 1]]
            cost_class = -prob[b][:, targets[b]["labels"]]
            # softplus(-o) = softplus(o) - o  =>  BCE cost = (sum softplus(o) - o.t) / P   (matcher.py:38-62)
            sp = F.softplus(o).sum(-1, keepdim=True)
            s = o.sigmoid()
            both = torch.cat([o, s], 0) @ t.T
            cost_mask = (sp - both[:Q]) / P
            cost_dice = 1 - (2 * both[Q:] + 1) / (s.sum(-1)[:, None] + t.sum(-1)[None, :] + 1)
            C = self.cost_mask * cost_mask + self.cost_class * cost_class + self.cost_dice * cost_dice
            costs.append(C.reshape(Q, -1))
        return costs

    @staticmethod
    def solve(cost_lists):
        """cost_lists: list (per call) of list (per image) of device tensors -> same nesting of
        (index_i, index_j) int64 CPU tensors.  One D2H transfer for everything."""
        flat = [c for cl in cost_lists for c in cl]
        sizes = [c.numel() for c in flat]
        host = torch.cat([c.reshape(-1).float() for c in flat]).cpu() if sum(sizes) > 0 else torch.zeros(0)
        res, off, k = [], 0, 0
        for cl in cost_lists:
            cur = []
            for c in cl:
                C = host[off:off + sizes[k]].view(c.shape)
                off += sizes[k]
                k += 1
                i, j = linear_sum_assignment(C.numpy())
                cur.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            res.append(cur)
        return res

    @torch.no_grad()
    def forward(self, outputs, targets):
        return self.solve([self.cost_matrices(outputs, targets)])[0]

    def __repr__(self, _repr_indent=4):
        body = [f"cost_class: {self.cost_class}", f"cost_mask: {self.cost_mask}", f"cost_dice: {self.cost_dice}"]
        return "\n".join(["Matcher " + self.__class__.__name__] + [" " * _repr_indent + line for line in body])
