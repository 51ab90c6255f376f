"""Point sampling on mask logits / ground-truth masks (detectron2.projects.point_rend
point_features.point_sample + get_uncertain_point_coords_with_randomness, used at
mask2former/modeling/criterion.py:164-182 and matcher.py:122-132).  Third-party semantics restated:
point_sample(x, c) = grid_sample(x, 2c-1, bilinear, zeros, align_corners=False)."""
import torch
import torch.nn.functional as F

from . import _rng


def point_sample(input, point_coords):
    """input [R,C,H,W]; point_coords [R,P,2] in [0,1]x[0,1] (x,y) -> [R,C,P]"""
    return F.grid_sample(input, 2.0 * point_coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros",
                         align_corners=False).squeeze(3)


def get_uncertain_point_coords_with_randomness(coarse_logits, num_points, oversample_ratio, importance_sample_ratio,
                                               tag):
    """uncertainty = -|logit| (criterion.py:73-87).  coarse_logits [R,1,H,W] -> coords [R,num_points,2]"""
    assert oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1
    R = coarse_logits.shape[0]
    dev = coarse_logits.device
    num_sampled = int(num_points * oversample_ratio)
    point_coords = _rng.rand(tag + "_over", (R, num_sampled, 2), dev)
    unc = -point_sample(coarse_logits, point_coords).abs()
    num_uncertain = int(importance_sample_ratio * num_points)
    num_random = num_points - num_uncertain
    idx = torch.topk(unc[:, 0, :], k=num_uncertain, dim=1)[1]
    idx = idx + num_sampled * torch.arange(R, dtype=torch.long, device=dev)[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(R, num_uncertain, 2)
    if num_random > 0:
        point_coords = torch.cat([point_coords, _rng.rand(tag + "_rand", (R, num_random, 2), dev)], dim=1)
    return point_coords
