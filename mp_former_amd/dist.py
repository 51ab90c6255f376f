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

    def __init__(self, groups, broadcast=True, wire_dtype=None):
        """``wire_dtype=torch.bfloat16``: the gradients cross the links as bf16 (half the bytes of a bucket) — each group owns a
        second flat buffer in that dtype, ``launch`` casts the fresh gradients straight into it, the collective SUMS it, and
        ``finish`` unpacks into the fp32 buffer the optimizer reads (`flat = wire * 1/world`, the division in fp32).  What the
        reference's hook point (train_net.py:307-322: the full-model clip wrapper sees the averaged gradients) receives is then a
        gradient with 8 mantissa bits per element and rank: tests/test_dist_cpu.py bounds the parameter drift against the fp32
        wire over 5 AdamW steps.  Default None = fp32 on the wire, bit-for-bit what DistributedDataParallel averages."""
        self.world = world_size()
        if wire_dtype in (None, torch.float32):
            wire_dtype = None
        elif wire_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("FlatGradSync: wire_dtype must be None / torch.float32, torch.bfloat16 or torch.float16")
        self.wire_dtype = wire_dtype
        self.record = False               # bench.py: events around the launches / the waits of finish() (timing())
        self._ev = None
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
            wire = torch.zeros(total, dtype=wire_dtype, device=dev) if wire_dtype is not None else None
            views, wviews = [], []
            for p, o in zip(params, offs):
                if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                    raise RuntimeError("FlatGradSync: dense parameters only")
                views.append(flat[o:o + p.numel()].as_strided(p.shape, p.stride()))       # the parameter's own layout
                if wire is not None:
                    wviews.append(wire[o:o + p.numel()].as_strided(p.shape, p.stride()))
            self.groups.append({"params": params, "flat": flat, "views": views, "wire": wire, "wviews": wviews, "handle": None,
                                "launched": False, "early": False})
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

    def bucket_bytes(self):
        """bytes each group's collective moves per rank (the wire buffer when there is one)"""
        bufs = [g["wire"] if g["wire"] is not None else g["flat"] for g in self.groups]
        return [int(b.numel() * b.element_size()) for b in bufs]

    @torch.no_grad()
    def launch(self, i):
        g = self.groups[i]
        if g["launched"]:
            return
        wire = g["wire"] is not None
        srcs, dsts = [], []
        for p, v in zip(g["params"], g["wviews"] if wire else g["views"]):
            if p.grad is None:
                v.zero_()                                                 # unused this step: contributes zeros
            elif wire or p.grad.data_ptr() != v.data_ptr():
                srcs.append(p.grad)
                dsts.append(v)
            p.grad = None            # (finish() puts the averaged view here; anything that shows up before is a late gradient)
        if dsts:
            torch._foreach_copy_(dsts, srcs)                              # (casts to the wire dtype where there is one)
        if self.record and self._ev is not None:
            self._ev["launch"][i].record()
        if distributed():
            buf = g["wire"] if wire else g["flat"]
            g["handle"] = dist.all_reduce(buf, op=dist.ReduceOp.SUM if wire else self._avg_op(), async_op=True)
        g["launched"] = True
        g["early"] = not self._in_finish

    _in_finish = False

    @torch.no_grad()
    def finish(self):
        self._in_finish = True
        try:
            for i, g in enumerate(self.groups):
                if g["launched"]:
                    late = [j for j, p in enumerate(g["params"]) if p.grad is not None]
                    if late:
                        raise RuntimeError(f"FlatGradSync: {len(late)} gradient(s) of group {i} arrived after its all-reduce was "
                                           "launched — those parameters belong in a later group")
                else:
                    self.launch(i)
        finally:
            self._in_finish = False
        rec = self.record and self._ev is not None
        if rec:
            self._ev["wait0"].record()
            self._ev["early"] = [bool(g["early"]) for g in self.groups]
        for i, g in enumerate(self.groups):
            if g["handle"] is not None:
                g["handle"].wait()
                g["handle"] = None
                if g["wire"] is None and self._avg_op() != dist.ReduceOp.AVG and self.world > 1:
                    g["flat"].mul_(1.0 / self.world)
            if rec:
                self._ev["done"][i].record()
            if g["wire"] is not None:                                     # unpack: fp32 <- wire, averaged in fp32
                g["flat"].copy_(g["wire"])
                if self.world > 1:
                    g["flat"].mul_(1.0 / self.world)
            for p, v in zip(g["params"], g["views"]):
                p.grad = v
            g["launched"] = False
        if rec:
            self._ev["wait1"].record()
            self._ev["n"] += 1

    # ---- diagnostics (bench.py at N > 1; never inside the timed region) ---------------------------------------------------
    def record_events(self, on=True):
        """Record events on the current stream at every launch, in front of finish()'s waits and behind each of them.  GPU only."""
        self.record = bool(on)
        if on:
            E = lambda: torch.cuda.Event(enable_timing=True)              # noqa: E731
            n = len(self.groups)
            self._ev = {"launch": [E() for _ in range(n)], "done": [E() for _ in range(n)], "wait0": E(), "wait1": E(), "n": 0,
                        "early": [False] * n}

    def timing(self):
        """{exposed_wait_ms, in_flight_ms[i], launched_early[i]} of the LAST recorded step (synchronises).  exposed_wait_ms = time the
        compute stream spends inside finish() from the first wait to the last unpack: what the step pays for the exchange after
        the overlap; in_flight_ms[i] = launch of bucket i -> its wait satisfied (includes the compute it hid under)."""
        if not self._ev or not self._ev["n"]:
            return None
        torch.cuda.synchronize()
        ev = self._ev
        return {"exposed_wait_ms": round(ev["wait0"].elapsed_time(ev["wait1"]), 4),
                "in_flight_ms": [round(ev["launch"][i].elapsed_time(ev["done"][i]), 4) for i in range(len(self.groups))],
                "launched_early": list(ev["early"])}

    @torch.no_grad()
    def standalone_allreduce_ms(self, reps=5):
        """Each bucket's collective alone on an idle device (events around a blocking all-reduce of the buffer the step sends), median
        of ``reps``: the figure an N-GPU line needs to tell a slow link from a lost overlap.  The buffers hold stale gradients —
        call between steps only (the next launch overwrites them)."""
        out = []
        for g in self.groups:
            buf = g["wire"] if g["wire"] is not None else g["flat"]
            scratch = torch.zeros_like(buf)
            ts = []
            for _ in range(reps + 1):
                if buf.is_cuda:
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if distributed():
                        dist.all_reduce(scratch)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                else:
                    import time
                    t0 = time.perf_counter()
                    if distributed():
                        dist.all_reduce(scratch)
                    ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[1:])
            out.append(round(ts[len(ts) // 2], 4))
        return out
