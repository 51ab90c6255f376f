"""The ground-truth masks of a batch as ONE [sum T_b, H, W] tensor.

The mask-piloted query setup (mask2former_transformer_decoder.py:968-998) and the matcher / criterion
(matcher.py:103-156, criterion.py:141-191) both want the per-image ``targets[b]["masks"]`` of
maskformer_model.py:281-299 stacked; each used to run its own concatenation pass over the same ~25 MB.
``stacked_masks`` keeps the last result and hands it out again while the SAME tensors (storage, shape, dtype,
version counter) are asked for."""
import torch

_last = None      # (key, inputs kept alive so their storage cannot be recycled under the key, result)


def stacked_masks(masks):
    """masks: list of [T_b, H, W] tensors of one dtype and spatial size -> torch.cat(masks) (cached for the step)."""
    global _last
    key = tuple((m.data_ptr(), tuple(m.shape), m.dtype, m._version) for m in masks)
    if _last is not None and _last[0] == key:
        return _last[2]
    out = torch.cat(list(masks)).contiguous()
    _last = (key, list(masks), out)
    return out
