"""The static part of the training step as HIP graphs.

At the fixed 1024 x 1024 crop of the COCO configs (coco_instance_new_baseline_dataset_mapper.py:60-64) everything from the
image to the pixel decoder's outputs has fixed shapes: R50 backbone and MSDeformAttn pixel decoder — ~850 of the step's
~1 250 kernel launches and ~14 of the ~25 ms the launch thread needs to enqueue a step (tools/host_regions.py), which is
MORE than the GPU needs to run it.  They are captured once (forward and backward, torch.cuda.make_graphed_callables: one
replay each per step) in three pieces,

    A  stem + res2 + res3      B  res4 + res5      C  pixel decoder (6 encoder layers, FPN, mask_features)

so that the gradient exchange keeps its two early launches: the tensor hook on res5 fires when C's backward is done (head
bucket), the one on res3 when B's is (res5 + res4 bucket) — bench.py / dist.FlatGradSync unchanged.  B and C are captured on
aliases of the upstream pieces' static OUTPUT buffers, so a forward replay reads its input where the previous replay left
it (no copy); the feature-map gradients between the pieces are copied into the downstream graph's static buffers
(125 MB per step, ~40 us).  The decoder, matcher and criterion depend on the ground truth of the batch (number of
mask-piloted queries, pair lists) and stay eager.

What made the capture possible: the item tables of the grouped launches travel as kernel arguments (mpf_upload_small)
instead of through pinned staging memory.  The in-library launch profiler records nothing inside a replay, so callers that
want per-kernel times (bench.py's roofline steps) run the eager path for those steps.
"""
import torch
from torch import nn

from . import _lib


class _StagesA(nn.Module):
    """stem + res2 + res3.  The wrapper holds ONLY its own stages: a graphed callable's static inputs are the wrapper's
    parameters, and with the whole backbone registered here (most of its parameters unused by this piece) the backward
    capture crashed inside hipStreamEndCapture on this ROCm build."""

    def __init__(self, bb):
        super().__init__()
        self.stem_conv, self.stem_norm, self.res2, self.res3 = bb.stem_conv, bb.stem_norm, bb.res2, bb.res3

    def forward(self, x):
        from .backbone import run_stages
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            f = run_stages(x, [("res2", self.res2), ("res3", self.res3)], (self.stem_conv, self.stem_norm))
        return f["res2"], f["res3"]


class _StagesB(nn.Module):
    def __init__(self, bb):
        super().__init__()
        self.res4, self.res5 = bb.res4, bb.res5

    def forward(self, res3):
        from .backbone import run_stages
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            f = run_stages(res3, [("res4", self.res4), ("res5", self.res5)])
        return f["res4"], f["res5"]


class _PixelDecoder(nn.Module):
    def __init__(self, pd):
        super().__init__()
        self.pd = pd

    def forward(self, res2, res3, res4, res5):
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            mf, _, ms = self.pd.forward_features({"res2": res2, "res3": res3, "res4": res4, "res5": res5})
        return (mf, *ms)


def _alias(t):
    """a leaf on the same memory (the downstream graph's static input IS the upstream graph's static output)"""
    return t.detach().requires_grad_(True)


class GraphedTrunk:
    """backbone + pixel decoder of a bench.TrainModel-like model (attributes ``backbone`` = backbone.ResNet50, ``head`` =
    head.MPFormerHead) as three graphed callables.  ``__call__(images)`` -> (feature dict, (mask_features, multi_scale))."""

    def __init__(self, backbone, pixel_decoder, sample_images, warmup=3, pieces="abc"):
        assert sample_images.is_cuda and not sample_images.requires_grad
        x = sample_images.contiguous(memory_format=torch.channels_last).clone()

        def mk(mod, args, on):           # pieces not listed run eagerly (debugging / partial capture)
            if not on:
                return mod
            # scratch buffers whose addresses the graph bakes in are private to it (_lib.workspace_scope)
            with _lib.workspace_scope(f"graph:{id(self)}:{type(mod).__name__}"):
                return torch.cuda.make_graphed_callables(mod, args, num_warmup_iters=warmup, allow_unused_input=True)

        self.a = mk(_StagesA(backbone), (x,), "a" in pieces)
        r2, r3 = self.a(x)
        self.b = mk(_StagesB(backbone), (_alias(r3),), "b" in pieces)
        r4, r5 = self.b(r3)
        self.c = mk(_PixelDecoder(pixel_decoder), (_alias(r2), _alias(r3), _alias(r4), _alias(r5)), "c" in pieces)
        self._x = x

    def __call__(self, images):
        r2, r3 = self.a(images.contiguous(memory_format=torch.channels_last))
        r4, r5 = self.b(r3)
        out = self.c(r2, r3, r4, r5)
        return {"res2": r2, "res3": r3, "res4": r4, "res5": r5}, (out[0], list(out[1:]))
