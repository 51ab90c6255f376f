"""Host -> device uploads that do not stall the launch thread.

A copy from PAGEABLE host memory blocks the calling thread until the stream reaches it (the runtime
stages it through its own pinned buffer), i.e. every ``tensor.to(device)`` / ``torch.tensor(...,
device=...)`` of host data is a device synchronisation in disguise — in the decoder forward that cost
the launch thread its whole head start over the GPU.  ``upload`` copies through PyTorch's caching
pinned-memory allocator with ``non_blocking=True`` instead (the allocator keeps the staging block
alive until the copy has run)."""
import numpy as np
import torch


# tables up to this size travel as kernel arguments (mpf_upload_small: <= 3 968 bytes per launch): no staging memory, one
# launch per 3.9 KB, and a HIP graph that captures the launch carries the bytes (the item tables of the grouped launches
# in the backbone / pixel decoder are functions of pointers that are fixed inside a graph's memory pool)
SMALL_BYTES = 2 * 3968


_NP2T = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32,
         np.dtype(np.uint8): torch.uint8, np.dtype(np.float64): torch.float64}


def upload(data, device, dtype=None):
    """numpy array / python list / CPU tensor -> device tensor, asynchronously."""
    if dtype is None and isinstance(data, np.ndarray) and data.dtype in _NP2T and isinstance(device, torch.device) \
            and device.type == "cuda":
        # the common case (item tables of the grouped launches): straight from the numpy buffer, no tensor wrappers
        nbytes = data.nbytes
        if 0 < nbytes <= SMALL_BYTES and nbytes % 4 == 0:
            from . import _lib
            a = data if data.flags.c_contiguous else np.ascontiguousarray(data)
            out = torch.empty(a.shape, dtype=_NP2T[a.dtype], device=device)
            with _lib.device_guard(device):
                code = _lib.lib().mpf_upload_small(a.__array_interface__["data"][0], out.data_ptr(), nbytes, _lib.stream_ptr(device))
            _lib.check(code, "mpf_upload_small")
            return out
    t = torch.from_numpy(data) if isinstance(data, np.ndarray) else torch.as_tensor(data)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device.type != "cpu":
        return t.to(device)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    nbytes = t.numel() * t.element_size()
    if 0 < nbytes <= SMALL_BYTES and nbytes % 4 == 0:
        from . import _lib
        t = t.contiguous()
        out = torch.empty(t.shape, dtype=t.dtype, device=device)
        with _lib.device_guard(device):
            code = _lib.lib().mpf_upload_small(t.data_ptr(), out.data_ptr(), nbytes, _lib.stream_ptr(device))
        _lib.check(code, "mpf_upload_small")
        return out
    return t.pin_memory().to(device, non_blocking=True)
