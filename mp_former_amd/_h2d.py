"""Host -> device uploads that do not stall the launch thread.

A copy from PAGEABLE host memory blocks the calling thread until the stream reaches it (the runtime
stages it through its own pinned buffer), i.e. every ``tensor.to(device)`` / ``torch.tensor(...,
device=...)`` of host data is a device synchronisation in disguise — in the decoder forward that cost
the launch thread its whole head start over the GPU.  ``upload`` copies through PyTorch's caching
pinned-memory allocator with ``non_blocking=True`` instead (the allocator keeps the staging block
alive until the copy has run)."""
import numpy as np
import torch


def upload(data, device, dtype=None):
    """numpy array / python list / CPU tensor -> device tensor, asynchronously."""
    t = torch.from_numpy(data) if isinstance(data, np.ndarray) else torch.as_tensor(data)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device.type != "cpu":
        return t.to(device)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)
