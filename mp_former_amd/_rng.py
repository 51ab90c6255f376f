"""Random draws of the hot path (label noise, matcher points, loss points), funnelled through one
place so tests can REPLAY the exact tensors the reference drew (SURVEY.md §7 "RNG parity").
Production draws come from torch's generator on the target device, batched into as few launches as
possible; under replay every (tag, shape) is served from the installed FIFO for that tag."""
import torch

_replay = None   # dict tag -> list[Tensor] (FIFO per tag) when a test installs a replay


def install_replay(by_tag):
    global _replay
    _replay = {k: list(v) for k, v in by_tag.items()} if by_tag is not None else None


def replaying():
    return _replay is not None


def remaining():
    return 0 if _replay is None else sum(len(v) for v in _replay.values())


def _take(tag, shape, device):
    t = _replay[tag].pop(0)
    assert tuple(t.shape) == tuple(shape), f"rng replay '{tag}': shape {tuple(t.shape)} != wanted {tuple(shape)}"
    return t.to(device)


def rand(tag, shape, device):
    if _replay is not None:
        return _take(tag, shape, device)
    return torch.rand(*shape, device=device)


def randint(tag, shape, high, device):
    if _replay is not None:
        return _take(tag, shape, device)
    return torch.randint(0, high, tuple(shape), device=device)


def rand_cat(parts, device):
    """parts: list of (tag, shape) whose shapes agree except in dim 0 -> their draws concatenated
    along dim 0 (one torch.rand in production)."""
    if _replay is not None:
        return torch.cat([_take(tag, shape, device) for tag, shape in parts], 0)
    if not parts:
        return torch.zeros(0, device=device)
    rows = sum(s[0] for _, s in parts)
    return torch.rand((rows,) + tuple(parts[0][1][1:]), device=device)
