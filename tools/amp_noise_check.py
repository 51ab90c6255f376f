import os, sys, json
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import fifo_to_tags, load_head_fixture
from mp_former_amd import _rng
from mp_former_amd.head import MPFormerHead
def rel(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
def run(name, factored=True, amp=True):
    dev = torch.device("cuda:0")
    z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
    h = MPFormerHead(num_classes=cfg["num_classes"], num_queries=cfg["num_queries"], enc_layers=cfg["enc_layers"], dec_layers=cfg["dec_layers"],
                     num_points=cfg["num_points"], noise_scale=cfg.get("noise_scale", 0.0), factored_masks=factored)
    h.pixel_decoder.load_state_dict(pp); h.predictor.load_state_dict(dp); h = h.to(dev).train()
    feats = {k: v.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True) for k, v in feats.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    _rng.install_replay(fifo_to_tags(replay, cfg, "dn_pred_logits" in z))
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            losses, _ = h(feats, targets)
            total = sum(losses.values())
        total.backward()
    finally:
        _rng.install_replay(None)
    errs = {k: rel(v.grad.detach().float().reshape(-1)[::7].cpu().numpy(), z[f"grad_feat_{k}_s7"]) for k, v in feats.items()}
    dg = dict(h.predictor.named_parameters())
    errs["query_feat"] = rel(dg["query_feat.weight"].grad.float().cpu().numpy(), z["grad_dec.query_feat.weight"])
    print(name, "factored" if factored else "dense", "amp" if amp else "fp32", os.environ.get("MPF_FUSED_DECODER", "1"), float(total), float(z["total_loss"]),
          {k: round(v, 4) for k, v in errs.items()})
for name in ("head_noise", "head_small"):
    run(name)
    run(name, factored=False)
    os.environ["MPF_FUSED_DECODER"] = "0"
    run(name)
    os.environ.pop("MPF_FUSED_DECODER")
    run(name, amp=False)
