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
    return ddp


class FlatGradSync:
    """Gradient averaging over the data-parallel ranks through a few FLAT buckets, without per-parameter hooks.

    ``groups``: lists of parameters in the order in which their backward COMPLETES (for the training step: the head's
    parameters, then the backbone's).  Each group owns one flat fp32 buffer; ``launch(i)`` copies the group's fresh
    gradients into it with one multi-tensor copy and starts ONE asynchronous all-reduce (RCCL: ``AVG``), ``finish()`` does
    the same for the groups not launched yet, waits, and leaves ``p.grad`` pointing at the averaged views (the optimizer
    reads them in place).  The caller launches an early group from a single autograd hook — bench.py: when the gradients
    of the backbone's feature maps are complete, so the head's 80 MB fly while the backbone back-propagates.

    Why not DistributedDataParallel here: its reducer runs a hook and a scaled copy kernel per parameter (318 of them) and
    bucket bookkeeping in the autograd thread — 1.6-3.5 ms per step at world size 1 on this step —
    to overlap an exchange that xGMI finishes in about a millisecond.  Two flat buckets keep the overlap where it pays
    (behind the backbone's backward) and drop the per-parameter work.  Parameters are broadcast from rank 0 once, as DDP
    does.  A gradient that arrives for a group AFTER its launch would be lost: ``finish`` checks and raises.

    Unused parameters: a parameter without a gradient on this rank contributes zeros and ends the step with the averaged
    view as its ``.grad`` (another rank may have used it), so the optimizer steps it (weight decay, moment decay) where the
    single-process run — whose ``.grad`` stays None — skips it.  DistributedDataParallel has the same semantics with
    ``find_unused_parameters``; the shipped training step uses every parameter on every rank (asserted by
    tests/test_head_gpu.py::test_head_config_A_runs_and_is_finite), so the two trajectories agree there.
    """

    def __init__(self, groups, broadcast=True):
        self.world = world_size()
        self.groups = []
        for params in groups:
            params = [p for p in params if p.requires_grad]
            if not params:
                continue
            dev = params[0].device
            offs, total = [], 0
            for p in params:
                if p.dtype != torch.float32 or p.device != dev:
                    raise RuntimeError("FlatGradSync: fp32 parameters on one device per group")
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64                     # 256-byte aligned slots
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            views = []
            for p, o in zip(params, offs):
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    raise RuntimeError("FlatGradSync: dense parameters only")
                views.append(flat[o:o + p.numel()].as_strided(p.shape, p.stride()))       # the parameter's own layout
            self.groups.append({"params": params, "flat": flat, "views": views, "handle": None, "launched": False})
        if broadcast and distributed():
            for g in self.groups:
                for p in g["params"]:
                    dist.broadcast(p.data, src=0)

    def _avg_op(self):
        # RCCL / NCCL average in the collective; gloo has no AVG: sum, then one scale per bucket.  One rank: the average IS the
        # sum, and RCCL's one-rank AVG would still run its pre-multiply pass over every bucket (0.21 ms per step for 176 MB)
        if self.world == 1:
            return dist.ReduceOp.SUM
        return dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM

    @torch.no_grad()
    def launch(self, i):
        g = self.groups[i]
        if g["launched"]:
            return
        srcs, dsts = [], []
        for p, v in zip(g["params"], g["views"]):
            if p.grad is None:
                v.zero_()                                                 # unused this step: contributes zeros
            elif p.grad.data_ptr() != v.data_ptr():
                srcs.append(p.grad)
                dsts.append(v)
            p.grad = None            # (finish() puts the averaged view here; anything that shows up before is a late gradient)
        if dsts:
            torch._foreach_copy_(dsts, srcs)
        if distributed():
            g["handle"] = dist.all_reduce(g["flat"], op=self._avg_op(), async_op=True)
        g["launched"] = True

    @torch.no_grad()
    def finish(self):
        for i, g in enumerate(self.groups):
            if g["launched"]:
                late = [j for j, p in enumerate(g["params"]) if p.grad is not None]
                if late:
                    raise RuntimeError(f"FlatGradSync: {len(late)} gradient(s) of group {i} arrived after its all-reduce was "
                                       "launched — those parameters belong in a later group")
            else:
                self.launch(i)
        for g in self.groups:
            if g["handle"] is not None:
                g["handle"].wait()
                g["handle"] = None
                if self._avg_op() != dist.ReduceOp.AVG and self.world > 1:
                    g["flat"].mul_(1.0 / self.world)
            for p, v in zip(g["params"], g["views"]):
                p.grad = v
            g["launched"] = False
