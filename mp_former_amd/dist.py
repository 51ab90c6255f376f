"""Data-parallel plumbing of the hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) — or gloo on CPU in the tests.  The path shards by image only; the
collectives are DDP's bucketed gradient all-reduce plus the scalar `num_masks` all-reduce of
mask2former/modeling/criterion.py:235-237."""
import os

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def distributed():
    return dist.is_available() and dist.is_initialized()


def init_from_env(backend=None, device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (torch.distributed.run sets them).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    force = os.environ.get("MPF_FORCE_DIST", "0") == "1" and "MASTER_ADDR" in os.environ   # 1-GPU test of the N>1 path
    if (world > 1 or force) and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl" and "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
            # ROCr reads the variable when HIP initialises: setting it here only helps if nothing has touched the GPU
            # yet (bench.py exports it before importing torch; torch.distributed.run children inherit the launcher's env)
            if torch.cuda.is_initialized():
                import warnings
                warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY=0 was not in the environment when HIP initialised: RCCL's "
                              "buffer sharing between processes needs it on this driver (hipIpcGetMemHandle: invalid "
                              "argument otherwise) — export it before the first torch.cuda call")
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"   # dmabuf IPC only on this driver
        kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def global_num_masks(local_count, device):
    """clamp(all_reduce_sum(#GT masks) / world_size, min=1) as a python float (criterion.py:224-237)."""
    ws = world_size()
    n = float(local_count)
    if distributed():
        t = torch.as_tensor([n], dtype=torch.float, device=device)
        dist.all_reduce(t)
        n = t.item()
    return max(n / ws, 1.0)


def global_num_masks_device(local_count, device):
    """The same value as a [1] tensor on ``device`` WITHOUT synchronising the host: the count goes up through
    pinned staging, the all-reduce is enqueued on the stream and the clamp runs on the device.  The reference
    (criterion.py:236-237) and ``global_num_masks`` call ``.item()``, which drains the stream once per step on
    every rank — the only host synchronisation the multi-GPU step would have left."""
    if device.type == "cuda":
        from ._h2d import upload
        t = upload([float(local_count)], device, torch.float32)
    else:
        t = torch.tensor([float(local_count)], dtype=torch.float32, device=device)
    if distributed():
        dist.all_reduce(t)
    return (t / world_size()).clamp_(min=1.0)


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (bench.py: step time = slowest rank)."""
    if not distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def wrap_ddp(model, device_ids=None):
    """DistributedDataParallel as Detectron2's DefaultTrainer builds it (broadcast_buffers=False) with
    gradients as bucket views; identity at world size 1."""
    if not distributed():
        return model
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=device_ids, broadcast_buffers=False,
                                                    gradient_as_bucket_view=True)
    if os.environ.get("MPF_DDP_BF16_ALLREDUCE", "0") == "1":
        # optional (off by default: the reference all-reduces fp32 gradients): halves the bytes each xGMI link
        # carries per step; gradients are rounded to bf16 for the exchange only
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
    return ddp
