"""Glue of the hot path: pixel decoder -> MP decoder -> criterion, i.e. what
MaskFormerHead.layers (mask2former/modeling/meta_arch/mask_former_head.py:118-121) and the training
branch of MaskFormer.forward (mask2former/maskformer_model.py:212-232) do between "backbone feature
dict in" and "weighted loss dict out".  Used by bench.py and the parity tests; a Detectron2 user
plugs the two decoders in through the registries instead."""
import torch
from torch import nn

from .criterion import SetCriterion
from .matcher import HungarianMatcher
from .pixel_decoder import MSDeformAttnPixelDecoder, ShapeSpec
from .transformer_decoder import MultiScaleMaskedTransformerDecoderMaskDN

R50_SHAPES = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}


def build_weight_dict(dec_layers, class_weight=2.0, mask_weight=5.0, dice_weight=5.0):
    """maskformer_model.py:121-132 (DEC_LAYERS counts the prediction on the learnable queries)."""
    wd = {"loss_ce": class_weight, "loss_mask": mask_weight, "loss_dice": dice_weight}
    wd.update({k + "_dn": v for k, v in list(wd.items())})
    aux = {}
    for i in range(dec_layers - 1):
        aux.update({k + f"_{i}": v for k, v in wd.items()})
    wd.update(aux)
    return wd


def prepare_targets(instances, padded_hw):
    """maskformer_model.py:281-299: per-image Detectron2 ``Instances`` (anything with ``image_size`` (h, w), ``gt_classes``,
    ``gt_masks`` [T, h, w] and optionally ``gt_boxes.tensor`` xyxy) -> the target dicts of the hot path: ground-truth masks
    zero-padded to the padded batch size (``images.tensor.shape[-2:]``), labels, boxes as normalised cxcywh (None without
    boxes, as the reference).  ONE allocation + copy per image; no device synchronisation."""
    h_pad, w_pad = int(padded_hw[0]), int(padded_hw[1])
    out = []
    for t in instances:
        h, w = t.image_size
        gm = t.gt_masks
        gm = gm.tensor if hasattr(gm, "tensor") else gm              # detectron2 BitMasks or a plain tensor
        if gm.shape[-2] > h_pad or gm.shape[-1] > w_pad:
            raise ValueError(f"ground-truth masks {tuple(gm.shape[-2:])} larger than the padded batch ({h_pad}, {w_pad})")
        padded = torch.zeros((gm.shape[0], h_pad, w_pad), dtype=gm.dtype, device=gm.device)
        padded[:, :gm.shape[1], :gm.shape[2]] = gm
        boxes = None
        if hasattr(t, "gt_boxes"):
            b = t.gt_boxes.tensor if hasattr(t.gt_boxes, "tensor") else t.gt_boxes
            x0, y0, x1, y1 = b.unbind(-1)
            cxcywh = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)            # util/box_ops.py:16-20
            boxes = cxcywh / torch.as_tensor([w, h, w, h], dtype=torch.float, device=b.device)
        out.append({"labels": t.gt_classes, "masks": padded, "boxes": boxes})
    return out


class MPFormerHead(nn.Module):
    """COCO-instance defaults (configs/coco/instance-segmentation/maskformer2_R50_bs16_50ep.yaml +
    run_50ep_no_noise_all_ly.sh): 100 queries, 80 classes, 6 encoder / 9 decoder layers, NUM_DN 1,
    NOISE_SCALE 0, DN_MODE points, ALL_LY_DN True, LB_NOISE_RATIO 0.2."""

    def __init__(self, num_classes=80, num_queries=100, enc_layers=6, dec_layers=9, num_points=12544,
                 feature_shapes=None, scalar=1, noise_scale=0.0, label_noise_ratio=0.2, hidden_dim=256, factored_masks=True):
        super().__init__()
        shapes = feature_shapes or R50_SHAPES
        self.pixel_decoder = MSDeformAttnPixelDecoder(
            {k: ShapeSpec(channels=c, stride=s) for k, (c, s) in shapes.items()},
            transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
            transformer_enc_layers=enc_layers, conv_dim=hidden_dim, mask_dim=hidden_dim, norm="GN",
            transformer_in_features=["res3", "res4", "res5"], common_stride=4)
        self.predictor = MultiScaleMaskedTransformerDecoderMaskDN(
            hidden_dim, True, num_classes=num_classes, hidden_dim=hidden_dim, num_queries=num_queries, nheads=8,
            dim_feedforward=2048, dec_layers=dec_layers, pre_norm=False, mask_dim=hidden_dim,
            enforce_input_project=False, dn_mode="points", head_dn=False, all_lys=True, dn_ratio=0.5,
            dn_label_noise_ratio=label_noise_ratio)
        # predictor and criterion are both ours: the mask predictions travel between them as their factors (mask_fused.py)
        self.predictor.factored_masks = bool(factored_masks)
        matcher = HungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=num_points)
        self.criterion = SetCriterion(num_classes, matcher=matcher, weight_dict=build_weight_dict(dec_layers + 1),
                                      eos_coef=0.1, losses=["labels", "masks"], num_points=num_points,
                                      oversample_ratio=3.0, importance_sample_ratio=0.75)
        self.scalar, self.noise_scale = scalar, noise_scale

    def forward(self, features, targets):
        """features: dict res2..res5; targets: list of {"labels","masks","boxes"} (prepare_targets
        format, maskformer_model.py:281-299).  Returns the WEIGHTED loss dict (:226-231)."""
        mask_features, _, multi_scale = self.pixel_decoder.forward_features(features)
        dn_args = {"tgt": targets, "scalar": self.scalar, "noise_scale": self.noise_scale}
        outputs = self.predictor(multi_scale, mask_features, None, dn_args)
        losses = self.criterion(outputs, targets)
        wd = self.criterion.weight_dict
        return {k: v * wd[k] for k, v in losses.items() if k in wd}, outputs

    def total_loss(self, features, targets):
        """Sum of the weighted losses without materialising the weighted dict (2 kernels instead of 120)."""
        mask_features, _, multi_scale = self.pixel_decoder.forward_features(features)
        dn_args = {"tgt": targets, "scalar": self.scalar, "noise_scale": self.noise_scale}
        outputs = self.predictor(multi_scale, mask_features, None, dn_args)
        return self.criterion.weighted_total(self.criterion(outputs, targets))
