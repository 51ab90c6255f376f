#!/usr/bin/env python3
"""bench.py — training images/sec of the MP-Former COCO-instance R50 step at 1024x1024 on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration on one synthetic batch already resident in HBM:
  R50 backbone (own folded-BN ResNet-50: MIOpen bf16 convolutions + a native bias / residual / ReLU epilogue)  ->  MSDeformAttn pixel decoder (fp32, native HIP
  deformable attention)  ->  masked-attention decoder with mask-piloted queries (bf16 autocast)  ->
  Hungarian matching + point-sampled CE / BCE / dice losses (60 terms)  ->  backward  ->  gradient
  all-reduce over RCCL (three flat buckets, two of them launched from tensor hooks under the backbone's backward,
  mp_former_amd.dist.FlatGradSync; MPF_GRAD_SYNC=ddp = torch DDP; N > 1)  ->  full-model grad-norm clip 0.01  ->  AdamW.
Per-GPU batch is fixed at 2 images (IMS_PER_BATCH 16 on 8 GPUs): weak scaling.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  "roofline":     achieved-vs-peak of the time-dominant native kernel family (gemm3: the encoder's / pixel decoder's fp32
                  GEMMs as split-bf16 MFMA products, ~40 % of the step) with the HBM-bound deformable-attention kernels,
                  the attention tiles and the fused mask kernels under "also" — all from HIP events recorded on the launch
                  stream around every launch of a few extra steps AFTER the timed region (which runs with the log off);
  "cpu_baseline": the oracle's CPU restatement of the hot path (pixel decoder + decoder +
                  criterion, forward + backward) timed on this box's host cores on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import sys
import time

# ROCr reads its flags when HIP initialises (the first torch.cuda call): the dmabuf-only IPC mode RCCL needs on this
# driver has to be in the environment BEFORE that, i.e. before anything below touches the GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# kernel arguments in device memory (this runtime's default; spelled out because the step is ~1 200 launches of mostly 3-10 us
# kernels: with HIP_FORCE_DEV_KERNARG=0 the same step takes 24.6 instead of 22.7 ms — tools/experiments/README.md)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PIXEL_MEAN = (123.675, 116.280, 103.530)
PIXEL_STD = (58.395, 57.120, 57.375)


def synth_targets(n, size, num_classes, gen, device):
    """T_b ~ U{1..20} random axis-aligned rectangles / ellipses as bool masks; size = side or (H, W)."""
    tg = []
    H, W = (size, size) if isinstance(size, int) else size
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for _ in range(n):
        T = int(torch.randint(1, 21, (1,), generator=gen))
        masks = torch.zeros(T, H, W, dtype=torch.bool)
        for t in range(T):
            cy, cx = (torch.rand(2, generator=gen) * torch.tensor([H, W], dtype=torch.float32)).tolist()
            hh, ww = (torch.rand(2, generator=gen) * torch.tensor([H, W], dtype=torch.float32) * 0.3 + 8).tolist()
            if torch.rand(1, generator=gen).item() < 0.5:
                masks[t] = ((yy - cy).abs() < hh) & ((xx - cx).abs() < ww)
            else:
                masks[t] = ((yy - cy) / hh) ** 2 + ((xx - cx) / ww) ** 2 < 1.0
            if not masks[t].any():
                masks[t, int(cy) % H, int(cx) % W] = True
        labels = torch.randint(0, num_classes, (T,), generator=gen)
        tg.append({"labels": labels.to(device), "masks": masks.to(device), "boxes": torch.zeros(T, 4, device=device)})
    return tg


def synth_batch(n, size, num_classes, seed, device):
    gen = torch.Generator().manual_seed(seed)
    img = torch.rand(n, 3, size, size, generator=gen) * 255.0
    mean = torch.tensor(PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(1, 3, 1, 1)
    img = ((img - mean) / std).to(device)
    return img, synth_targets(n, size, num_classes, gen, device)


class TrainModel(torch.nn.Module):
    """backbone + head + criterion: forward returns the summed weighted loss (what DDP wraps)."""

    def __init__(self, num_classes=80, num_queries=100):
        super().__init__()
        from mp_former_amd.backbone import ResNet50
        from mp_former_amd.head import MPFormerHead
        self.backbone = ResNet50()
        self.head = MPFormerHead(num_classes=num_classes, num_queries=num_queries)

    # feature name -> callable, run when the backward pass reaches that feature map, i.e. when every parameter that was used
    # AFTER it in the forward pass has its final gradient (FlatGradSync.launch): "res5" = the head is done (the head's nodes
    # were created last, so autograd runs all of them before the first backbone node), "res3" = res5 and res4 are done too
    grad_ready_hooks = None

    def forward(self, images, targets):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats = self.backbone(images.contiguous(memory_format=torch.channels_last))
            if self.grad_ready_hooks and torch.is_grad_enabled():
                for name, cb in self.grad_ready_hooks.items():
                    if feats[name].requires_grad:
                        feats[name].register_hook(lambda g, cb=cb: cb())     # (returns None: the gradient is unchanged)
            return self.head.total_loss(feats, targets)


# BASELINE.json configs as head-only workloads (--workload X --head-only): the segmentation head on synthetic backbone features
# of the configuration's channel counts (SURVEY.md 8(d) allows synthetic features; the backbone is outside the path), per-GPU
# batch 2 = IMS_PER_BATCH 16 on 8 GPUs in all four configs.  R50: configs/coco/*/maskformer2_R50_bs16_50ep.yaml; Swin-B / ADE20K
# 640: configs/ade20k/semantic-segmentation/swin/maskformer2_swin_base_384_bs16_160k_res640.yaml; Swin-L / Cityscapes 1024 x 2048,
# 200 queries: configs/cityscapes/panoptic-segmentation/swin/maskformer2_swin_large_IN21k_384_bs16_90k.yaml:18.
WORKLOADS = {
    "B": dict(name="COCO-instance R50", hw=(1024, 1024), chans=(256, 512, 1024, 2048), classes=80, queries=100),
    "C": dict(name="COCO-panoptic R50", hw=(1024, 1024), chans=(256, 512, 1024, 2048), classes=133, queries=100),
    "D": dict(name="ADE20K-semantic Swin-B", hw=(640, 640), chans=(128, 256, 512, 1024), classes=150, queries=100),
    "E": dict(name="Cityscapes-panoptic Swin-L", hw=(1024, 2048), chans=(192, 384, 768, 1536), classes=19, queries=200),
}


class HeadOnlyModel(torch.nn.Module):
    """pixel decoder + MP decoder + criterion on a dict of backbone feature maps (bf16 channel-last planes, what a bf16 backbone
    hands over under autocast); the features require gradients, so the backward does everything the full step's head does."""
    grad_ready_hooks = None

    def __init__(self, wl):
        super().__init__()
        from mp_former_amd.head import MPFormerHead
        self.shapes = {f"res{i + 2}": (c, st) for i, (c, st) in enumerate(zip(wl["chans"], (4, 8, 16, 32)))}
        self.head = MPFormerHead(num_classes=wl["classes"], num_queries=wl["queries"], feature_shapes=self.shapes)

    def forward(self, feats, targets):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return self.head.total_loss(feats, targets)


def synth_features(n, hw, shapes, seed, device):
    gen = torch.Generator().manual_seed(seed)
    H, W = hw
    return {k: torch.randn(n, H // st, W // st, c, generator=gen).to(device=device, dtype=torch.bfloat16).permute(0, 3, 1, 2).requires_grad_(True)
            for k, (c, st) in shapes.items()}


def _gemm3_traffic_ratios():
    """{shape: HBM bytes from the PMC counters / algorithmic bytes} of the fp16 x 2 TN kernels the routing picks
    (profiles/r05_gemm3_traffic.json from tools/pmc_gemm3_traffic.sh: gemm3_tn3 for N = 256, gemm3_ws for N = 1024), or None"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm3_traffic.json")))
    if not fs:
        return None
    f = fs[-1]
    d = json.load(open(f))
    return {k: v["fp16x2"]["traffic_over_algorithmic"] for k, v in d.get("shapes", {}).items() if "fp16x2" in v}


def grad_sync_groups(model):
    """Three flat buckets in the order in which their gradients complete: the head (80 MB), the backbone's res5 + res4
    (88 MB, 94 % of the backbone) and the rest (res3, res2, stem: 6 MB) — only the last one is exchanged after backward()."""
    if not hasattr(model, "backbone"):                  # --head-only
        return [list(model.head.parameters())]
    bb = model.backbone
    late = list(bb.res5.parameters()) + list(bb.res4.parameters())
    ids = {id(p) for p in late}
    early = [p for p in bb.parameters() if id(p) not in ids]
    return [list(model.head.parameters()), late, early]


def build_optimizer(model):
    """AdamW 1e-4, wd 0.05, backbone lr x0.1, no decay on norms / embeddings (train_net.py:259-337)."""
    buckets = {}
    norm_types = (torch.nn.LayerNorm, torch.nn.GroupNorm, torch.nn.BatchNorm2d)
    seen = set()
    for mname, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            lr, wd = 1e-4, 0.05
            if "backbone" in mname:
                lr *= 0.1
            if isinstance(module, norm_types) or isinstance(module, torch.nn.Embedding) or "level_embed" in pname:
                wd = 0.0
            buckets.setdefault((lr, wd), []).append(p)   # 4 groups -> 4 fused multi-tensor launches
    groups = [{"params": ps, "lr": lr, "weight_decay": wd} for (lr, wd), ps in buckets.items()]
    # full-model clip (CLIP_VALUE 0.01, train_net.py:316-320) + AdamW in three native launches
    from mp_former_amd.optim import ClipAdamW
    return ClipAdamW(groups, lr=1e-4, max_norm=0.01)


def _cpu_baseline_one(size, timed_steps, threads):
    from oracle import head_ref as O
    from mp_former_amd.head import MPFormerHead
    torch.manual_seed(0)
    ref = MPFormerHead()   # parameter container only (CPU tensors); the math below is the oracle's
    pp = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.pixel_decoder.state_dict().items()}
    dp = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.predictor.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    feats = {k: torch.randn(1, c, size // s, size // s, generator=gen).requires_grad_(True)
             for k, (c, s) in {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}.items()}
    targets = synth_targets(1, size, 80, gen, "cpu")
    cfg = {"num_queries": 100, "num_classes": 80}
    times = []
    for it in range(1 + timed_steps):
        t0 = time.time()
        total, _ = O.head_step(pp, dp, feats, targets, cfg)
        total.backward()
        if it > 0:
            times.append(time.time() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_baseline(size, threads=None):
    """Oracle (CPU restatement, kind 'port') of the hot path on this box's host cores:
    pixel decoder + MP decoder + criterion, forward + backward, fp32, N=1; config B (1024x1024, the metric's
    configuration) is `value`, config A (256x256, the reference's own CPU-runnable case) is reported beside it.
    1 warm-up + 5 timed steps each, median."""
    # the oracle is many small/medium fp32 ops: beyond ~16 threads OpenMP fork/join dominates (measured on
    # the 256-CPU GPU box at 1024x1024: 8 thr 6.6 s/step, 16 thr 4.3, 32 thr 5.4, 256 thr > 300), so cap
    # the thread count and report the number actually used
    cores = threads or min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    med_a, n_a = _cpu_baseline_one(256, 5, cores)
    med_b, n_b = _cpu_baseline_one(size, 5, cores)
    return {"value": round(1.0 / med_b, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"hot path only (pixel decoder + MP decoder + criterion, fwd+bwd, fp32), {size}x{size}, N=1, "
                      f"1 warm-up + {n_b} timed steps, median {med_b:.2f} s/step, torch {torch.get_num_threads()} threads",
            "config_A_256x256": {"value": round(1.0 / med_a, 3), "unit": "images/sec",
                                 "sample": f"same path at 256x256, N=1, 1 warm-up + {n_a} timed steps, median {med_a:.3f} s/step"}}


def _sha256(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start one FRESH process per GPU the way the reference's driver does
    (train_net.py:399-412 -> detectron2 `launch(main, num_gpus, ...)`), here as a `torch.distributed.run` child of this process
    with the same arguments, relay its output (rank 0's single JSON line) and return its exit status.  Called before
    anything in this process has initialised HIP — the parent never touches the GPU and never replaces itself (no exec).
    MPF_BENCH_LAUNCH_SCRIPT (tests only): the script the ranks run instead of this file (tests/_bench_world2_child.py maps
    both ranks to the one GPU of the test box)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.environ.get("MPF_BENCH_LAUNCH_SCRIPT", os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + sys.argv[1:]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:                      # (stderr is inherited; stdout relayed line by line)
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra steps after the timed region with the launch log on (roofline block)")
    ap.add_argument("--msda-offsets", choices=["init", "trained"], default="init",
                    help="sampling_offsets of the six encoder layers: the module's initial pattern (ms_deform_attn.py:66-74), or "
                         "'trained' = that pattern + N(0, 3 px) noise written into the biases and N(0, 0.02) weights (per-query scatter), "
                         "the regime of a mid-training step")
    ap.add_argument("--trained-steps", type=int, default=10, help="timed steps of the trained-offsets second number (0 = skip)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="B",
                    help="BASELINE.json configuration (B = the metric's; C, D, E need --head-only: their backbones are outside the path)")
    ap.add_argument("--head-only", action="store_true",
                    help="time the segmentation head alone on synthetic backbone features of the workload's channel counts")
    ap.add_argument("--grad-wire", choices=["fp32", "bf16"], default="fp32",
                    help="dtype of the gradient buckets on the links (N > 1, FlatGradSync): fp32 = what DistributedDataParallel averages (default); "
                         "bf16 = half the bytes, unpacked and averaged in fp32 (SURVEY.md 8 f4)")
    ap.add_argument("--head-steps", type=int, default=10,
                    help="N = 1, full step only: timed HEAD-ONLY steps after everything else (config.head_only; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-size", type=int, default=1024)
    a = ap.parse_args()
    if a.workload != "B" and not a.head_only:
        raise SystemExit("--workload C / D / E are head-only workloads (their backbones are outside the path): add --head-only")
    wl = WORKLOADS[a.workload]
    hw = wl["hw"] if (a.head_only and a.workload != "B") else (a.size, a.size)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))            # (nothing has touched the GPU yet: children are fresh processes)
    assert world == a.gpus, f"WORLD_SIZE {world} != --gpus {a.gpus}"
    dev_index = local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    from mp_former_amd import dist as mdist
    mdist.init_from_env("nccl", dev)                     # nccl == RCCL on ROCm

    from mp_former_amd import _lib, _miopen
    _lib.lib()   # fail loudly if the native library is missing
    _miopen.use_shipped_find_db(check_version=True)      # tuned MIOpen solver choice (private copy; logs a version mismatch)

    from mp_former_amd import dropin
    dropin.configure_training_process(single_thread_autograd=True)          # backward on the launch thread (one device per process)

    torch.manual_seed(rank)
    if os.environ.get("MPF_CONV_FIND", "0") == "1":      # let MIOpen time its solvers per conv shape (slow warm-up)
        torch.backends.cudnn.benchmark = True
    if a.head_only:
        model = HeadOnlyModel(wl).to(dev).train()
    else:
        model = TrainModel().to(dev).train()
        model.backbone.to(memory_format=torch.channels_last)

    def trained_like_offsets():
        """sampling offsets of a mid-training model: per-query scatter of sigma ~ 3 px (weights N(0, 0.18) on the O(1) query
        features) around the initial pattern moved by N(0, 1 px) (biases).  A freshly built MSDeformAttn has zero offset weights,
        i.e. every query of a head looks the same way: the regime in which bounding-box designs are at their best."""
        g_ = torch.Generator(device="cpu").manual_seed(1234)          # the same on every rank: replicas stay identical
        with torch.no_grad():
            for n_, p_ in model.head.pixel_decoder.named_parameters():
                if n_.endswith("sampling_offsets.bias"):
                    p_.add_(torch.randn(p_.shape, generator=g_).to(dev))
                elif n_.endswith("sampling_offsets.weight"):
                    p_.copy_((torch.randn(p_.shape, generator=g_) * 0.18).to(dev))

    if a.msda_offsets == "trained":
        trained_like_offsets()
    # gradient exchange: three flat buckets (head | res5 + res4 | rest of the backbone), the first two launched from two
    # tensor hooks while the backbone back-propagates (mp_former_amd.dist.FlatGradSync); MPF_GRAD_SYNC=ddp selects torch's
    # DistributedDataParallel (per-parameter hooks, 25 MB buckets)
    sync = None
    if mdist.distributed() and os.environ.get("MPF_GRAD_SYNC", "flat") == "flat":
        sync = mdist.FlatGradSync(grad_sync_groups(model), wire_dtype=torch.bfloat16 if a.grad_wire == "bf16" else None)
        if not a.head_only:
            model.grad_ready_hooks = {"res5": lambda: sync.launch(0), "res3": lambda: sync.launch(1)}
        ddp = model
    else:
        ddp = mdist.wrap_ddp(model, [dev_index])
    opt = build_optimizer(model)
    if a.head_only:
        batches = []
        for i in range(4):
            g_ = torch.Generator().manual_seed(1000 * rank + i)
            batches.append((synth_features(a.batch, hw, model.shapes, 1000 * rank + i, dev), synth_targets(a.batch, hw, wl["classes"], g_, dev)))
    else:
        batches = [synth_batch(a.batch, a.size, 80, 1000 * rank + i, dev) for i in range(4)]
    def step(i):
        images, targets = batches[i % len(batches)]
        opt.zero_grad(set_to_none=True)
        loss = ddp(images, targets)
        loss.backward()
        if sync is not None:
            sync.finish()
        opt.step()
        return loss

    def barrier():
        if mdist.distributed():
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    barrier()
    # ---- the metric: K steps, launch profiler OFF (no event records inside the timed region) -------------------------
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    dt = mdist.max_over_ranks(dt, dev)
    final_loss = float(loss.detach())

    # ---- roofline block: a few MORE steps with the in-library launch log on (HIP events recorded on the launch stream
    # around every native kernel; outside the timed region, so the headline does not pay for the event records) --------
    P = a.profile_steps
    _lib.profile_enable(True)
    for i in range(P):
        step(a.warmup + a.steps + i)
    barrier()

    def prof(name):
        n, ms, by = _lib.profile_get(name)
        return n, ms, by, _lib.profile_get_flops(name)

    n_push, ms_push, _, _ = prof("msda_bwd_bin")      # round 4: destination-side backward = bin + tile launches per call
    n_pull, ms_pull, _, _ = prof("msda_bwd_tile")
    n_f, ms_f, by_f, _ = prof("msda_fwd")
    attn = {k: prof(k) for k in ("attn_fwd_kernel", "attn_bwd_kv_kernel", "attn_bwd_q_kernel")}
    # gemm3 family: per group (launches, ms, fp32-equivalent flops 2 M N K, MFMA flops actually issued, algorithmic bytes).  An
    # fp32 product is three MFMA products in the fp16 x 2 form ("h2" in the launch label) and where one operand is a bf16
    # matrix, six in the bf16 x 3 form.  "<128, bf16>" (one product: the decoder's K / V projections) is not part of the family.
    def fam(all_names, three_names=()):
        n = ms = fl = by = fl3 = 0.0
        for nm in all_names:
            n_, ms_, by_, fl_ = prof(nm)
            n, ms, fl, by = n + n_, ms + ms_, fl + fl_, by + by_
        for nm in three_names:
            fl3 += prof(nm)[3]
        return int(n), ms, fl, 3.0 * fl3 + 6.0 * (fl - fl3), by

    g_tn = fam(["gemm3_tn_kernel"], ["gemm3_tn_kernel<h2", "gemm3_tn_kernel<a16>"])
    g_nt = fam(["gemm3_nt_kernel<128>", "gemm3_nt_kernel<96>", "gemm3_nt_kernel<a16>", "gemm3_nt_kernel<b16>"],
               ["gemm3_nt_kernel<128>h2", "gemm3_nt_kernel<96>h2", "gemm3_nt_kernel<a16>", "gemm3_nt_kernel<b16>"])
    g_ng = fam(["gemm3_nt_group_kernel"], ["gemm3_nt_group_kernel<h2>"])      # the four plain gradients of an encoder layer, one launch
    g_cv = fam(["gemm3_conv_kernel"], ["gemm3_conv_kernel<h2>"])              # 3x3 FPN convolution (forward + input gradient), K = 9 Cin
    g_cw = fam(["gemm3_nt_kernel<conv3x3"], ["gemm3_nt_kernel<conv3x3 h2>"])  # ... and its weight gradient
    n_am, ms_am, by_am, _ = prof("amax_kernel")
    fused = {k: prof(k) for k in ("match_cost_fused_kernel", "pair_planes_fwd_kernel", "pair_planes_dfeat_kernel", "pair_planes_dembed_kernel")}
    _lib.profile_enable(False)
    # ---- gradient exchange diagnostics (N > 1, flat buckets): two more steps with events at every bucket launch and around
    # finish()'s waits, then each bucket's collective alone on the idle device — enough to tell a slow link from a lost
    # overlap in a SCALE line without a profiler
    grad_sync_info = None
    if sync is not None:
        sync.record_events(True)
        for i in range(2):
            step(a.warmup + a.steps + P + i)
        tim = sync.timing()
        sync.record_events(False)
        alone = sync.standalone_allreduce_ms()
        bb_ = sync.bucket_bytes()
        w_ = max(world, 1)
        grad_sync_info = {"buckets_mb": [round(b / 1e6, 2) for b in bb_], "wire_dtype": a.grad_wire,
                          "allreduce_ms": alone,
                          # ring all-reduce bus bandwidth: 2 (n - 1) / n x bytes / time
                          "allreduce_busbw_GBps": [round(2.0 * (w_ - 1) / w_ * b / (t * 1e-3) / 1e9, 1) if t > 0 else None for b, t in zip(bb_, alone)],
                          "exposed_wait_ms": tim["exposed_wait_ms"], "in_flight_ms": tim["in_flight_ms"],
                          "launched_under_backbone": bool(all(tim["launched_early"][:-1])) if len(tim["launched_early"]) > 1 else False,
                          "launched_early": tim["launched_early"],
                          "what": "allreduce_ms = each bucket's collective alone (median of 5, idle device); exposed_wait_ms = compute-stream time "
                                  "from finish()'s first wait to its last unpack on a step; in_flight_ms = bucket launch -> wait satisfied (includes "
                                  "the backbone backward it ran under)"}
    S_tok = sum((hw[0] // s) * (hw[1] // s) for s in (8, 16, 32))
    n_b = n_pull                                   # one bin + one tile launch per MSDA backward call
    ms_b = ms_push + ms_pull
    by_b = 1344.0 * 4 * S_tok * a.batch * n_b      # algorithmic bytes: SURVEY.md §8(d), fp32, per call
    # HBM traffic per call from rocprofv3 PMC passes (tools/pmc_msda.sh: FETCH_SIZE + WRITE_SIZE in separate passes, KiB
    # units, FETCH_SIZE calibrated on a known-byte copy — see the file) on the same shape; counters cannot be read from
    # inside the process, so the file carries the hash of the kernel source it was measured on and is REFUSED (traffic =
    # null) when the kernels have changed since
    traffic, traffic_note = None, "no PMC file for this shape"
    import glob
    pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_msda_bwd_pmc_configB_N2.json")))       # the newest round's file
    pmc_file = pmc_files[-1] if pmc_files else ""
    if hw == (1024, 1024) and a.batch == 2 and pmc_file:
        pmc = json.load(open(pmc_file))
        src = os.path.join(ROOT, "mp_former_amd", "csrc", "msda_block.hip")
        if pmc.get("source_sha256") == _sha256(src):
            traffic = pmc["hbm_bytes_per_call"]
            traffic_note = pmc.get("note", "")
        else:
            traffic_note = os.path.basename(pmc_file) + " is older than csrc/msda_block.hip: refused"

    # ---- the hot path alone, in the driver's own record: the SAME model, the segmentation head on the
    # detached feature maps of one cached backbone forward per batch (the features require gradients, so the head's backward
    # does all the work it does in the full step), clip + AdamW over the head's parameters.  After every other timed region.
    head_leg = None
    if world == 1 and not a.head_only and a.head_steps > 0:
        cached = []
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for images, targets in batches:
                f_ = model.backbone(images.contiguous(memory_format=torch.channels_last))
                cached.append(({k: v.detach().clone().requires_grad_(True) for k, v in f_.items()}, targets))
        saved_hooks, model.grad_ready_hooks = model.grad_ready_hooks, None

        def head_step(i):
            feats, targets = cached[i % len(cached)]
            opt.zero_grad(set_to_none=True)
            for v in feats.values():
                v.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                l_ = model.head.total_loss(feats, targets)
            l_.backward()
            opt.step()                                  # (parameters without a gradient — the backbone's — are skipped)
            return l_

        for i in range(3):
            head_step(i)
        barrier()
        t2 = time.perf_counter()
        for i in range(a.head_steps):
            head_step(3 + i)
        barrier()
        h_ms = (time.perf_counter() - t2) / a.head_steps * 1e3
        model.grad_ready_hooks = saved_hooks
        head_leg = {"ms_per_step": round(h_ms, 2), "images_per_sec": round(a.batch / h_ms * 1e3, 2), "steps": a.head_steps,
                    "msda_offsets": a.msda_offsets,
                    "what": "hot path only: pixel decoder + MP decoder + matching + 60 losses + backward + clip + AdamW of the head, on the "
                            "detached bf16 feature maps of a cached backbone forward of the same model and batches"}

    # second number: the same step with trained-like sampling offsets (the headline above is iteration 0 of a
    # freshly initialised model); 3 warm-up + 10 timed steps, launch log off
    trained_ms = None
    if a.msda_offsets == "init" and a.trained_steps > 0:
        trained_like_offsets()
        for i in range(3):
            step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(a.trained_steps):
            step(3 + i)
        barrier()
        trained_ms = mdist.max_over_ranks(time.perf_counter() - t1, dev) / a.trained_steps * 1e3

    # replicas check (on by default up to 8 ranks, MPF_CHECK_SYNC=0 turns it off): the ranks' parameters must have stayed
    # identical through the averaged updates — outside every timed region
    sync_spread = None
    if mdist.distributed() and os.environ.get("MPF_CHECK_SYNC", "1" if world <= 8 else "0") == "1":
        ps_ = list(model.parameters())
        cs = torch.stack([p.detach().double().sum() for p in ps_])                 # signed sum per parameter tensor
        scale = torch.stack([p.detach().double().abs().sum() for p in ps_]) + 1e-30
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)
        allcs = torch.stack(allcs)
        sync_spread = float(((allcs.max(0).values - allcs.min(0).values) / scale).max())

    # what the process group actually was (a SCALE line then proves RCCL saw N ranks): backend, ranks, RCCL's version
    pg_info = None
    if mdist.distributed():
        pg_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "grad_sync": os.environ.get("MPF_GRAD_SYNC", "flat")}
        if pg_info["backend"] == "nccl":
            pg_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    if rank == 0:
        ips = a.batch * world * a.steps / dt
        achieved = by_b / (ms_b * 1e-3) / 1e9 if ms_b > 0 else 0.0

        def attn_entry(name):
            n, ms, by, fl = attn[name]
            if not n or ms <= 0:
                return {"kernel": name, "launches": 0}
            tf = fl / (ms * 1e-3) / 1e12
            return {"kernel": name, "launches_per_step": n / max(P, 1), "ms_per_step": round(ms / max(P, 1), 3), "bound": "hbm/L2 stream of K, V, mask",
                    "mfma_tflops": round(tf, 2), "mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS, "mfma_frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 5),
                    "kv_mask_GBps": round(by / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}

        FP32_MFMA_PEAK_TFLOPS = 157.0            # native fp32 MFMA peak (MI355X_MICROARCH.md): what a true-fp32 GEMM could reach

        RIDGE = MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9)        # 312.5 flop/B

        def gemm_entry(name, grp):
            """`frac` is ALGORITHMIC and taken against the roofline that binds the group: algorithmic flop/B = 2 M N K over
            4 (M K + M N) + weight planes; below the ridge (312 flop/B: the encoder's Linear layers, 64-115) the group is HBM-bound and
            frac = algorithmic bytes / time / 8 TB/s; above it (the 3 x 3 FPN convolution as a GEMM with K = 9 Cin: ~570) it is
            MFMA-bound and frac = fp32-equivalent flops / time / the dense bf16 MFMA peak.  The MFMA products actually issued (three
            per fp32 product in the fp16 x 2 form) are separate keys either way."""
            n, ms, fl, issued, by = grp
            if not n or ms <= 0:
                return {"kernel": name, "launches": 0}
            sec = ms * 1e-3
            hf = by / sec / 1e9 / HBM_PEAK_GBPS
            mf = fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS
            mfma_bound = fl / max(by, 1.0) > RIDGE
            e = {"kernel": name, "launches_per_step": n / max(P, 1), "ms_per_step": round(ms / max(P, 1), 3),
                 "algorithmic_flop_per_byte": round(fl / max(by, 1.0), 1)}
            if mfma_bound:
                e.update({"bound": "mfma", "achieved": round(fl / sec / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mf, 4),
                          "hbm_frac": round(hf, 4)})
            else:
                e.update({"bound": "hbm", "achieved": round(by / sec / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hf, 4)})
            e.update({"algorithmic_tflops_fp32": round(fl / sec / 1e12, 1), "algorithmic_mfma_frac": round(mf, 4),
                      "issued_mfma_tflops": round(issued / sec / 1e12, 1), "issued_mfma_frac": round(issued / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                      "x_native_fp32_mfma_peak_157": round(fl / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3)})
            return e

        # the family figure covers its HBM-bound members (TN, NT, grouped NT: the encoder's Linear layers and their gradients); the
        # two 3 x 3 convolution groups are MFMA-bound on algorithmic terms and are reported on their own under "also"
        gemm_parts = [g_tn, g_nt, g_ng]
        g_ms_all = sum(x[1] for x in (g_tn, g_nt, g_ng, g_cv, g_cw))
        g_n = sum(x[0] for x in gemm_parts)
        g_ms = sum(x[1] for x in gemm_parts)
        g_fl = sum(x[2] for x in gemm_parts)
        g_issued = sum(x[3] for x in gemm_parts)
        g_by = sum(x[4] for x in gemm_parts)
        g_sec = max(g_ms * 1e-3, 1e-12)
        g_mf, g_hf = g_issued / g_sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, g_by / g_sec / 1e9 / HBM_PEAK_GBPS

        def fused_entry(name, unit_bytes_note):
            n, ms, by, fl = fused[name]
            if not n or ms <= 0:
                return {"kernel": name, "launches": 0}
            e = {"kernel": name, "launches_per_step": n / max(P, 1), "ms_per_step": round(ms / max(P, 1), 3),
                 "GBps": round(by / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "bytes": unit_bytes_note}
            if fl > 0:
                e["mfma_tflops"] = round(fl / (ms * 1e-3) / 1e12, 2)
                e["mfma_frac"] = round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 5)
            return e

        fwd_alg = 800.0 * 4 * S_tok * a.batch * n_f            # the forward's own algorithmic bytes
        if a.head_only:
            metric = "training images/sec, segmentation head only (synthetic backbone features), %s %dx%d" % (wl["name"], hw[0], hw[1])
            workload = ("%s, %d queries, %d classes, %dx%d, HEAD ONLY: MSDeformAttn pixel decoder (6 layers) + MP masked decoder (9 layers, "
                        "NUM_DN 1) + Hungarian matching + 60 losses + backward + clip + AdamW of the head, on synthetic bf16 backbone "
                        "features with channels %s" % (wl["name"], wl["queries"], wl["classes"], hw[0], hw[1], list(wl["chans"])))
        else:
            metric = "training images/sec COCO-instance R50 1024x1024"
            workload = ("COCO-instance R50, 100 queries, 80 classes, %dx%d: full train step = R50 backbone "
                        "+ MSDeformAttn pixel decoder (6 layers) + MP masked decoder (9 layers, NUM_DN 1) "
                        "+ Hungarian matching + 60 losses + backward + grad all-reduce + clip + AdamW" % (a.size, a.size))
        out = {
            "metric": metric,
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 (backbone, decoder) + f32 (pixel decoder, losses), as the reference's AMP",
            "data": "synthetic",
            "config": {"workload": workload, "baseline_config": a.workload, "head_only_workload": bool(a.head_only),
                       "global_batch": a.batch * world, "per_gpu_batch": a.batch, "parallelism": f"dp{world}",
                       "tokens_per_image_S": S_tok, "final_loss": round(final_loss, 4),
                       "roofline_steps": P, "msda_offsets": a.msda_offsets,
                       "autograd_threads": "backward on the launch thread",
                       "process_group": pg_info},
            # the time-dominant native kernel family: the fp32 GEMMs of the pixel decoder as SIX bf16 MFMA products per fp32
            # product (three 8-bit-mantissa planes per operand, terms >= 2^-16 kept).  achieved = bf16 MFMA flops actually
            # issued per second; peak = the dense bf16 MFMA peak
            "roofline": {"kernel": "gemm3 family (gemm3_tn / gemm3_nt / gemm3_nt_group: the fp32 Linear layers of the pixel decoder and their gradients as fp16 x 2 (three) or bf16 x 3 (six) MFMA products; the MFMA-bound 3 x 3 convolution groups are listed under also)",
                         # ALGORITHMIC fraction: the family's shapes are HBM-bound on algorithmic terms (64-115 fp32 flop/B,
                         # ridge 312), so achieved = algorithmic bytes / time against 8 TB/s; the MFMA products issued are separate keys
                         "bound": "hbm", "achieved": round(g_by / g_sec / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(g_hf, 4), "traffic": None,
                         "traffic_note": "algorithmic bytes = 4 (M K + M N) + 4 | 6 N K per launch (activations in, result out, weight planes); a family of "
                                         "~90 launches of different shapes has no per-launch PMC figure — the calibrated FETCH_SIZE / WRITE_SIZE of its TN "
                                         "kernel on the three encoder shapes (tools/pmc_gemm3_traffic.sh) is in tn_traffic_over_algorithmic",
                         "tn_traffic_over_algorithmic": _gemm3_traffic_ratios(),
                         "launches_per_step": g_n / max(P, 1), "ms_per_step": round(g_ms / max(P, 1), 3),
                         "ms_per_step_incl_conv3x3": round(g_ms_all / max(P, 1), 3),
                         "avg_us": round(g_ms * 1e3 / max(g_n, 1), 1),
                         "issued_mfma_tflops": round(g_issued / g_sec / 1e12, 1), "issued_mfma_frac": round(g_mf, 4),
                         "algorithmic_tflops_fp32": round(g_fl / g_sec / 1e12, 1),
                         "algorithmic_mfma_frac": round(g_fl / g_sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                         "x_native_fp32_mfma_peak_157": round(g_fl / g_sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3),
                         "amax_passes": {"launches_per_step": n_am / max(P, 1), "ms_per_step": round(ms_am / max(P, 1), 3)},
                         "also": [
                             gemm_entry("gemm3_tn_kernel", g_tn), gemm_entry("gemm3_nt_kernel", g_nt),
                             gemm_entry("gemm3_nt_group_kernel", g_ng), gemm_entry("gemm3_conv_kernel", g_cv),
                             gemm_entry("gemm3_nt_kernel<conv3x3>", g_cw),
                             # the deformable-sampling kernels against the HBM roofline (north-star)
                             {"kernel": "MSDA backward (msda_bwd_bin_kernel + msda_bwd_tile_kernel: destination-side, atomics-free, no bounding boxes)",
                              "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                              "launches": n_b, "avg_us": round(ms_b * 1e3 / max(n_b, 1), 1),
                              "avg_us_bin": round(ms_push * 1e3 / max(n_push, 1), 1), "avg_us_tile": round(ms_pull * 1e3 / max(n_pull, 1), 1),
                              "algorithmic_bytes_per_launch": round(by_b / max(n_b, 1))},
                             {"kernel": "msda_fwd_block_kernel", "launches": n_f, "avg_us": round(ms_f * 1e3 / max(n_f, 1), 1),
                              "bound": "hbm", "achieved": round(fwd_alg / (ms_f * 1e-3) / 1e9 if ms_f > 0 else 0.0, 1), "unit": "GB/s",
                              "frac": round(fwd_alg / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms_f > 0 else 0.0, 4),
                              "algorithmic_bytes_per_launch": round(fwd_alg / max(n_f, 1)),
                              "frac_incl_loc_attn_written": round(by_f / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms_f > 0 else 0.0, 4),
                              "note": "frac = SURVEY 8(d)'s 800*e*S*N bytes; frac_incl_loc_attn_written adds the 288*e*Lq*N bytes of sampling "
                                      "locations / attention weights this launch also writes for the backward (the msda_prep pass fused in)"},
                             # masked cross- / self-attention on bf16 MFMA tiles: MFMA rate against the gfx950 peak AND the
                             # K / V / mask stream rate (bound by the latter at ~120 queries: 4 MFMAs per 32 keys)
                             attn_entry("attn_fwd_kernel"), attn_entry("attn_bwd_kv_kernel"), attn_entry("attn_bwd_q_kernel"),
                             # the mask predictions from their factors (csrc/mask_fused.hip): no [N, 10 Qtot, H/4, W/4] tensor
                             fused_entry("match_cost_fused_kernel", "corner rows gathered: 2048 B per point and (output, image) + target samples"),
                             fused_entry("pair_planes_fwd_kernel", "features read once per image"),
                             fused_entry("pair_planes_dfeat_kernel", "features-gradient written + gradient planes read"),
                             fused_entry("pair_planes_dembed_kernel", "features + gradient planes read")]},
            "cpu_baseline": None,
        }
        if trained_ms is not None:
            out["config"]["trained_like_offsets"] = {"ms_per_step": round(trained_ms, 2), "images_per_sec": round(a.batch * world / trained_ms * 1e3, 2),
                                                     "steps": a.trained_steps,
                                                     "what": "same step, sampling offsets scattered per query (sigma ~ 3 px): offset weights N(0, 0.18), biases + N(0, 1 px)"}
        if sync_spread is not None:
            out["config"]["param_sync_spread"] = sync_spread
        if grad_sync_info is not None:
            out["config"]["grad_sync"] = grad_sync_info
        if head_leg is not None:
            out["config"]["head_only"] = head_leg
        out["config"]["fp32_gemm"] = ("fp16 x 2 split: two pieces per operand, three MFMA products, power-of-two scale from the operand's "
                                      "largest magnitude; error vs fp64 <= the library fp32 GEMM's (tests/test_gemm3_gpu.py)")
        out["config"]["miopen_find_db"] = "mismatch (MIOpen ignored the shipped db)" if _miopen.db_mismatch() else "shipped"
        if world == 1 and not a.no_cpu_baseline and not a.head_only:      # (the oracle's step is the config-B / config-A head)
            out["cpu_baseline"] = cpu_baseline(a.cpu_baseline_size)
        print(json.dumps(out), flush=True)
    if mdist.distributed():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
