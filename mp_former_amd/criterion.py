"""SetCriterion — mirror of mask2former/modeling/criterion.py (:21-65 dice / sigmoid-CE,
:73-87 uncertainty, :90-304 SetCriterion incl. the mask-piloted `*_dn` losses).  Same constructor,
same `forward(outputs, targets) -> dict` with the reference's 6 x (1 + #aux) keys, same `weight_dict`
attribute read by the caller (maskformer_model.py:226-231).

Scheduling difference (results identical for identical random draws): the Hungarian matchings of the
final and all auxiliary outputs are computed FIRST, with a single device->host copy of all cost
matrices, then the losses; the reference interleaves matcher and losses per output and blocks on a
`.cpu()` per image per output (matcher.py:149) plus `num_masks.item()` (criterion.py:237).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .matcher import GTMasks
from .point_sample import MaskLossSums, map_rows, uncertain_point_coords


def dice_loss(inputs, targets, num_masks: float):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(-1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    loss = 1 - (numerator + 1) / (denominator + 1)
    return loss.sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks: float):
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.mean(1).sum() / num_masks


def _world_size():
    return torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio, dn_no_lb=False):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses, self.dn_no_lb = eos_coef, losses, dn_no_lb
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio = num_points, oversample_ratio
        self.importance_sample_ratio = importance_sample_ratio
        self._gt = None   # per-forward cache of the batch's byte GT masks

    # ---- losses ----------------------------------------------------------------------------------
    def loss_labels(self, outputs, targets, indices, num_masks, tag=None):
        src_logits = outputs["pred_logits"].float()
        idx = self._get_src_permutation_idx(indices, src_logits.device)
        target_classes_o = torch.cat([t["labels"][J.to(t["labels"].device)] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64, device=src_logits.device)
        target_classes[idx] = target_classes_o
        loss_ce = F.cross_entropy(src_logits.transpose(1, 2), target_classes, self.empty_weight)
        return {"loss_ce": loss_ce}

    def loss_masks(self, outputs, targets, indices, num_masks, tag="loss"):
        """criterion.py:141-191 on the fused native kernels: the matched prediction maps and the byte
        ground-truth masks are sampled in place (no gather / float copies)."""
        src_masks = outputs["pred_masks"]
        dev = src_masks.device
        gt = self._gt if self._gt is not None else GTMasks(targets)
        b_idx, q_idx = self._get_src_permutation_idx(indices, dev)
        pred_rows = map_rows(src_masks, (b_idx, q_idx))
        off = torch.tensor(gt.offsets[:-1], dtype=torch.int64)
        gt_rows = torch.cat([J + off[i] for i, (_, J) in enumerate(indices)]).to(dev).to(torch.int32)
        with torch.no_grad():
            coords = uncertain_point_coords(src_masks, pred_rows, self.num_points, self.oversample_ratio,
                                            self.importance_sample_ratio, tag)
        sums = MaskLossSums.apply(src_masks, pred_rows, gt.u8, gt_rows, coords)
        P = self.num_points
        loss_mask = (sums[:, 0] / P).sum() / num_masks
        loss_dice = (1 - (2 * sums[:, 1] + 1) / (sums[:, 2] + sums[:, 3] + 1)).sum() / num_masks
        return {"loss_mask": loss_mask, "loss_dice": loss_dice}

    @staticmethod
    def _get_src_permutation_idx(indices, device):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx.to(device), src_idx.to(device)

    def get_loss(self, loss, outputs, targets, indices, num_masks, tag):
        loss_map = {"labels": self.loss_labels, "masks": self.loss_masks}
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, targets, indices, num_masks, tag)

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, outputs, targets):
        outputs_without_aux = {k: v for k, v in outputs.items() if k != "aux_outputs" and k != "dn_out"}
        dn_out = outputs["dn_out"]
        aux = outputs.get("aux_outputs", [])
        device = outputs["pred_logits"].device
        losses = {}
        num_masks = sum(len(t["labels"]) for t in targets)
        ws = _world_size()
        if ws > 1:   # criterion.py:235-237
            nm = torch.as_tensor([num_masks], dtype=torch.float, device=device)
            torch.distributed.all_reduce(nm)
            num_masks = nm.item()
        num_masks = max(num_masks / ws, 1.0)

        # all matchings first: one D2H copy for (1 + #aux) x N cost matrices
        self._gt = GTMasks(targets)
        cost_lists = [self.matcher.cost_matrices(outputs_without_aux, targets, "match", self._gt)]
        for i, a in enumerate(aux):
            cost_lists.append(self.matcher.cost_matrices(a, targets, f"match_{i}", self._gt))
        all_indices = self.matcher.solve(cost_lists)

        use_dn = bool(self.training and dn_out)
        if use_dn:
            dn_args = dn_out["dn_args"]
            scalar = dn_args["pad_size"] // dn_args["max_num"]
            dn_indices = []
            for t in targets:    # criterion.py:249-258: slot j of every DN group <-> GT j
                n = len(t["labels"])
                tt = torch.arange(n).unsqueeze(0).repeat(scalar, 1)
                oi = (torch.arange(scalar) * dn_args["max_num"]).unsqueeze(1) + tt
                dn_indices.append((oi.flatten().long(), tt.flatten().long()))

        def block(out, dn, indices, suffix):
            for loss in self.losses:
                l_dict = self.get_loss(loss, out, targets, indices, num_masks, "loss" + suffix)
                losses.update({k + suffix: v for k, v in l_dict.items()})
            if use_dn:
                for loss in self.losses:
                    l_dict = self.get_loss(loss, dn, targets, dn_indices, num_masks * scalar, "loss_dn" + suffix)
                    losses.update({k + "_dn" + suffix: v for k, v in l_dict.items()})
            else:
                z = torch.as_tensor(0.0, device=device)
                losses.update({"loss_mask_dn" + suffix: z, "loss_dice_dn" + suffix: z, "loss_ce_dn" + suffix: z})

        block(outputs_without_aux, {k: v for k, v in dn_out.items() if k != "aux_outputs"} if use_dn else None,
              all_indices[0], "")
        for i, a in enumerate(aux):
            block(a, dn_out["aux_outputs"][i] if use_dn else None, all_indices[i + 1], f"_{i}")
        self._gt = None
        if self.dn_no_lb:
            losses = {k: v for k, v in losses.items() if not k.startswith("loss_ce_dn")}
        return losses

    def __repr__(self):
        body = [f"matcher: {self.matcher.__repr__(_repr_indent=8)}", f"losses: {self.losses}",
                f"weight_dict: {self.weight_dict}", f"num_classes: {self.num_classes}", f"eos_coef: {self.eos_coef}",
                f"num_points: {self.num_points}", f"oversample_ratio: {self.oversample_ratio}",
                f"importance_sample_ratio: {self.importance_sample_ratio}"]
        return "\n".join(["Criterion " + self.__class__.__name__] + [" " * 4 + line for line in body])
