"""Import harness for the READ-ONLY reference at /root/reference (this container only).

Used ONLY by tests/golden/make_golden.py to generate golden vectors and, optionally, by
local (non-GPU-box) cross-checks.  Nothing of the reference travels with the repo: the
outputs are small .npz fixtures.  Recipe follows SURVEY.md Appendix C.

The reference needs detectron2 / fvcore / torchvision, none of which is installed; the hot-path
files only use a handful of symbols from them, which are stubbed here with their documented
semantics (third-party, un-vendored: see DESIGN.md "Oracle / third-party arithmetic").
"""
import importlib
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REF = os.environ.get("MPF_REFERENCE", "/root/reference")
M2F = os.path.join(REF, "mask2former")
PKG = "_mpf_ref"  # fake top-level package name standing in for `mask2former`

_done = False


def available():
    return os.path.isdir(M2F)


# ---- third-party semantics (restated; detectron2.projects.point_rend.point_features) ----------
def point_sample(input, point_coords, **kwargs):
    add_dim = False
    if point_coords.dim() == 3:
        add_dim = True
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    if add_dim:
        output = output.squeeze(3)
    return output


def get_uncertain_point_coords_with_randomness(coarse_logits, uncertainty_func, num_points,
                                               oversample_ratio, importance_sample_ratio):
    assert oversample_ratio >= 1
    assert 0 <= importance_sample_ratio <= 1
    num_boxes = coarse_logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    point_coords = torch.rand(num_boxes, num_sampled, 2, device=coarse_logits.device)
    point_logits = point_sample(coarse_logits, point_coords, align_corners=False)
    point_uncertainties = uncertainty_func(point_logits)
    num_uncertain_points = int(importance_sample_ratio * num_points)
    num_random_points = num_points - num_uncertain_points
    idx = torch.topk(point_uncertainties[:, 0, :], k=num_uncertain_points, dim=1)[1]
    shift = num_sampled * torch.arange(num_boxes, dtype=torch.long, device=coarse_logits.device)
    idx += shift[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(num_boxes, num_uncertain_points, 2)
    if num_random_points > 0:
        point_coords = torch.cat(
            [point_coords, torch.rand(num_boxes, num_random_points, 2, device=coarse_logits.device)],
            dim=1)
    return point_coords


# ---- detectron2.layers stand-ins --------------------------------------------------------------
class Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    assert norm == "GN", norm
    return nn.GroupNorm(32, out_channels)


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def configurable(init_func=None, *, from_config=None):
    # construct with explicit kwargs; no cfg plumbing needed for fixtures
    if init_func is not None:
        return init_func
    return lambda f: f


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def setup():
    """Install the stubs and the fake parent packages (idempotent)."""
    global _done
    if _done:
        return
    assert available(), f"reference not found at {REF}"
    _mod("detectron2")
    _mod("detectron2.config", configurable=configurable)
    _mod("detectron2.layers", Conv2d=Conv2d, ShapeSpec=ShapeSpec, get_norm=get_norm, DeformConv=object)
    reg = Registry("SEM_SEG_HEADS")
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=reg)
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=Registry)
    _mod("detectron2.utils.comm", get_world_size=lambda: 1)
    _mod("detectron2.projects")
    _mod("detectron2.projects.point_rend")
    _mod("detectron2.projects.point_rend.point_features", point_sample=point_sample,
         get_uncertain_point_coords_with_randomness=get_uncertain_point_coords_with_randomness)
    _mod("fvcore")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=c2_xavier_fill, c2_msra_fill=c2_msra_fill)
    _mod("fvcore.nn", weight_init=wi)
    if "torchvision" not in sys.modules:
        _mod("torchvision", _is_tracing=lambda: False, __version__="0.0")

    def _raise(*a, **k):
        raise RuntimeError("MultiScaleDeformableAttention stub: native op unavailable (CPU golden run)")

    _mod("MultiScaleDeformableAttention", ms_deform_attn_forward=_raise, ms_deform_attn_backward=_raise)

    _pkg(PKG, M2F)
    _pkg(PKG + ".modeling", os.path.join(M2F, "modeling"))
    _pkg(PKG + ".modeling.transformer_decoder", os.path.join(M2F, "modeling", "transformer_decoder"))
    _pkg(PKG + ".modeling.pixel_decoder", os.path.join(M2F, "modeling", "pixel_decoder"))
    _pkg(PKG + ".modeling.pixel_decoder.ops", os.path.join(M2F, "modeling", "pixel_decoder", "ops"))
    _pkg(PKG + ".utils", os.path.join(M2F, "utils"))

    # neutralise hard-coded device moves of the MP path (DEC:984-985,1029,1052; CRIT:251-257)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_to = torch.Tensor.to

    def _to(self, *args, **kwargs):
        args = tuple(a for a in args if not (isinstance(a, str) and a.startswith("cuda")))
        if isinstance(kwargs.get("device"), str) and kwargs["device"].startswith("cuda"):
            kwargs.pop("device")
        if not args and not kwargs:
            return self
        return _orig_to(self, *args, **kwargs)

    torch.Tensor.to = _to
    _done = True


def load(name):
    """load('modeling.pixel_decoder.msdeformattn') -> reference module object."""
    setup()
    return importlib.import_module(PKG + "." + name)


def msda_func():
    return load("modeling.pixel_decoder.ops.functions.ms_deform_attn_func")


def msda_module():
    return load("modeling.pixel_decoder.ops.modules.ms_deform_attn")


def pixel_decoder():
    return load("modeling.pixel_decoder.msdeformattn")


def decoder():
    return load("modeling.transformer_decoder.mask2former_transformer_decoder")


def criterion():
    return load("modeling.criterion")


def matcher():
    return load("modeling.matcher")


class RandCapture:
    """Record (and optionally replay) tensors drawn by torch.rand / rand_like / randint_like."""

    def __init__(self, replay=None):
        self.log = []
        self.spec = []
        self.replay = list(replay) if replay is not None else None
        self._orig = {}

    def _wrap(self, name):
        orig = getattr(torch, name)

        def f(*a, **k):
            if self.replay is not None:
                t = self.replay.pop(0)
                return t.clone()
            t = orig(*a, **k)
            self.log.append((name, t.detach().clone()))
            # how to draw the same tensor again from the same generator state (conftest.regenerate_draws): rand / rand_like fill
            # an empty tensor of that shape with uniform_(); randint_like(t, low, high) fills with random_(low, high)
            if name == "randint_like":
                lo, hi = (a[1], a[2]) if len(a) >= 3 else (k.get("low", 0), a[1] if len(a) >= 2 else k["high"])
                self.spec.append([name, list(t.shape), str(t.dtype).split(".")[-1], int(lo), int(hi)])
            else:
                self.spec.append([name, list(t.shape), str(t.dtype).split(".")[-1]])
            return t
        return orig, f

    def __enter__(self):
        for n in ("rand", "rand_like", "randint_like"):
            orig, f = self._wrap(n)
            self._orig[n] = orig
            setattr(torch, n, f)
        return self

    def __exit__(self, *exc):
        for n, o in self._orig.items():
            setattr(torch, n, o)
        return False
