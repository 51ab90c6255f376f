"""Deterministic (closed-form) parameters and inputs shared by make_golden.py (reference side) and
the tests (oracle / HIP side), so that multi-megabyte state dicts never have to be stored: both
sides call det_state_dict() on their own module / key list and get bit-identical tensors."""
import zlib

import numpy as np
import torch


def closed_form(n, salt):
    i = np.arange(n, dtype=np.float64)
    x = np.sin(i * 12.9898 + (salt % 9973) * 78.233) * 43758.5453
    return x - np.floor(x)


def det_tensor(name, shape, kind=None):
    """Values in a range that keeps a randomly wired network numerically tame."""
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = closed_form(n, zlib.crc32(name.encode())).reshape(shape) * 2.0 - 1.0   # U(-1,1)
    leaf = name.split(".")[-1]
    if name.endswith("sampling_offsets.bias"):
        return torch.from_numpy(np.ascontiguousarray(u * 3.0)).float()   # offsets of a few pixels
    if kind is None:
        if leaf == "bias" or leaf == "in_proj_bias":
            kind = "bias"
        elif ("norm" in name and leaf == "weight") or name.endswith(".1.weight") and "input_proj" in name:
            kind = "gain"
        else:
            kind = "weight"
    if kind == "bias":
        v = 0.05 * u
    elif kind == "gain":
        v = 1.0 + 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        v = u * (1.5 / np.sqrt(max(fan_in, 1)))
        if "sampling_offsets.weight" in name:
            v = v * 0.3          # keep offsets at a few pixels
        if name.endswith("embed.weight") or name.endswith("level_embed") or "query_feat" in name or "label_enc" in name:
            v = u * 0.5
    return torch.from_numpy(np.ascontiguousarray(v)).float()


def det_state_dict(shapes, prefix=""):
    """shapes: {key: shape} (e.g. from module.state_dict())."""
    out = {}
    for k, s in shapes.items():
        out[k] = det_tensor(prefix + k, s)
    return out


def det_features(N, size, channels=(("res2", 256, 4), ("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))):
    feats = {}
    for name, c, stride in channels:
        h = size // stride
        u = closed_form(N * c * h * h, zlib.crc32(("feat." + name).encode())).reshape(N, c, h, h)
        feats[name] = torch.from_numpy((u * 2.0 - 1.0)).float()
    return feats


def det_targets(N, size, counts, num_classes):
    """Axis-aligned rectangles as bool masks [T,size,size], labels, boxes (cxcywh, only len() is used)."""
    tg = []
    for b in range(N):
        T = counts[b]
        masks = torch.zeros(T, size, size, dtype=torch.bool)
        labels = torch.zeros(T, dtype=torch.int64)
        for t in range(T):
            u = closed_form(4, 1000 * b + t)
            y0 = int(u[0] * size * 0.6)
            x0 = int(u[1] * size * 0.6)
            hh = max(4, int(u[2] * size * 0.4))
            ww = max(4, int(u[3] * size * 0.4))
            masks[t, y0:y0 + hh, x0:x0 + ww] = True
            labels[t] = (7 * b + 3 * t) % num_classes
        tg.append({"labels": labels, "masks": masks, "boxes": torch.zeros(T, 4)})
    return tg
