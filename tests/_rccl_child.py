"""Child process of tests/test_rccl_gpu.py: three optimizer steps of the head on one GPU, either plain or through the
N > 1 plumbing (process group on backend nccl = RCCL, DDP with gradient bucket views, device-side num_masks all-reduce)
at world size 1.  Prints one JSON line with the per-step losses and a parameter checksum."""
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before HIP initialises (see mp_former_amd/dist.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    mode = sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from mp_former_amd import dist as mdist
    from mp_former_amd.head import MPFormerHead
    from mp_former_amd.optim import ClipAdamW
    if mode in ("ddp", "flat", "flat-bf16"):
        rank, world = mdist.init_from_env("nccl", dev)
        assert (rank, world) == (0, 1) and mdist.distributed(), "process group was not initialised"
        assert torch.distributed.get_backend() == "nccl"
    else:
        assert not mdist.distributed()
    torch.manual_seed(0)
    size = 256

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.head = MPFormerHead(num_classes=80, num_queries=50, enc_layers=2, dec_layers=3, num_points=1024)

        def forward(self, feats, targets):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.head.total_loss(feats, targets)

    model = M().to(dev).train()
    sync = None
    if mode in ("flat", "flat-bf16"):   # two flat buckets (what bench.py does at N > 1); "-bf16": bf16 on the wire (RCCL sums bf16)
        ps = [p for p in model.parameters() if p.requires_grad]
        sync = mdist.FlatGradSync([ps[:len(ps) // 2], ps[len(ps) // 2:]], wire_dtype=torch.bfloat16 if mode.endswith("bf16") else None)
        sync.record_events(True)
        ddp = model
    else:
        ddp = mdist.wrap_ddp(model, [0])
    if mode == "ddp":
        assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    opt = ClipAdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 1e-4, "weight_decay": 0.05}],
                    lr=1e-4, max_norm=0.01)
    g = torch.Generator().manual_seed(1)
    feats = {k: torch.randn(2, c, size // s, size // s, generator=g).to(dev)
             for k, (c, s) in {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}.items()}
    targets = []
    for b in range(2):
        T = 3 + 2 * b
        m = torch.zeros(T, size, size, dtype=torch.bool)
        for t in range(T):
            m[t, 20 * t:20 * t + 60, 30 * t:30 * t + 90] = True
        targets.append({"labels": (torch.arange(T) * 3 % 80).to(dev), "masks": m.to(dev), "boxes": torch.zeros(T, 4, device=dev)})
    losses = []
    for step in range(3):
        torch.manual_seed(100 + step)                     # identical point draws in both modes
        opt.zero_grad(set_to_none=True)
        loss = ddp(feats, targets)
        loss.backward()
        if sync is not None:
            sync.finish()
            assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                       for g in sync.groups for p, v in zip(g["params"], g["views"]))
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    checksum = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    extra = {}
    if sync is not None:        # the diagnostics bench.py prints at N > 1, on real RCCL
        extra = {"timing": sync.timing(), "allreduce_ms": sync.standalone_allreduce_ms(reps=3), "bucket_bytes": sync.bucket_bytes()}
    print("RESULT " + json.dumps({"mode": mode, "losses": losses, "checksum": checksum, **extra}), flush=True)
    if mdist.distributed():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
