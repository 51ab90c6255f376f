"""Hungarian matcher — mirror of mask2former/modeling/matcher.py (:15-62 cost functions, :70-179
HungarianMatcher).  Cost matrices are built on the GPU; the assignment is SciPy's
linear_sum_assignment on the host exactly as in the reference (:149-151), but ALL cost matrices of a
call (every image, and in `match_many` every decoder layer) travel in ONE device->host copy instead
of one blocking `.cpu()` per image per layer — assignments are unchanged.
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from . import _rng
from .point_sample import point_sample


def batch_dice_cost(inputs, targets):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * torch.einsum("nc,mc->nm", inputs, targets)
    denominator = inputs.sum(-1)[:, None] + targets.sum(-1)[None, :]
    return 1 - (numerator + 1) / (denominator + 1)


def batch_sigmoid_ce_cost(inputs, targets):
    hw = inputs.shape[1]
    pos = F.softplus(-inputs)     # BCE-with-logits against all-ones
    neg = F.softplus(inputs)      # ... against all-zeros
    loss = torch.einsum("nc,mc->nm", pos, targets) + torch.einsum("nc,mc->nm", neg, (1 - targets))
    return loss / hw


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.num_points = num_points

    @torch.no_grad()
    def cost_matrices(self, outputs, targets, tag="match"):
        """[C_b of shape [Q, T_b]] on the device (matcher.py:103-148)."""
        bs, num_queries = outputs["pred_logits"].shape[:2]
        costs = []
        for b in range(bs):
            out_prob = outputs["pred_logits"][b].float().softmax(-1)
            tgt_ids = targets[b]["labels"]
            cost_class = -out_prob[:, tgt_ids]
            out_mask = outputs["pred_masks"][b]
            tgt_mask = targets[b]["masks"].to(out_mask)
            point_coords = _rng.rand(tag, (1, self.num_points, 2), out_mask.device)
            tgt_pts = point_sample(tgt_mask[:, None], point_coords.repeat(tgt_mask.shape[0], 1, 1)).squeeze(1)
            out_pts = point_sample(out_mask[:, None], point_coords.repeat(out_mask.shape[0], 1, 1)).squeeze(1)
            with torch.autocast(device_type="cuda", enabled=False):
                out_pts, tgt_pts = out_pts.float(), tgt_pts.float()
                cost_mask = batch_sigmoid_ce_cost(out_pts, tgt_pts)
                cost_dice = batch_dice_cost(out_pts, tgt_pts)
            C = self.cost_mask * cost_mask + self.cost_class * cost_class + self.cost_dice * cost_dice
            costs.append(C.reshape(num_queries, -1))
        return costs

    @staticmethod
    def solve(cost_lists):
        """cost_lists: list (per call) of list (per image) of device tensors -> same nesting of
        (index_i, index_j) int64 CPU tensors.  One D2H transfer for everything."""
        flat = [c for cl in cost_lists for c in cl]
        sizes = [c.numel() for c in flat]
        if sum(sizes) > 0:
            host = torch.cat([c.reshape(-1).float() for c in flat]).cpu()
        else:
            host = torch.zeros(0)
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
