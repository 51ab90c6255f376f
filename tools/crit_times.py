#!/usr/bin/env python3
"""Fine-grained wall timing (with syncs) of the criterion's stages at config B, N=2."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mp_former_amd.matcher import GTMasks

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.TrainModel().to(dev).train()
images, targets = bench.synth_batch(2, 1024, 80, 0, dev)
crit = model.head.criterion
with torch.autocast("cuda", dtype=torch.bfloat16):
    feats = model.backbone(images)
    mf, _, ms = model.head.pixel_decoder.forward_features(feats)
    out = model.head.predictor(ms, mf, None, {"tgt": targets, "scalar": 1, "noise_scale": 0.0})
print("T per image:", [len(t["labels"]) for t in targets], "Qtot", out["pred_masks"].shape, out["dn_out"]["pred_masks"].shape)

def T(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r

t, gt = T(lambda: GTMasks(targets)); print(f"GTMasks build          {t:7.3f} ms")
outs = [{k: v for k, v in out.items() if k in ("pred_logits", "pred_masks")}] + out["aux_outputs"]
t, costs = T(lambda: [crit.matcher.cost_matrices(o, targets, "m", gt) for o in outs]); print(f"cost matrices x10      {t:7.3f} ms")
t, idx = T(lambda: crit.matcher.solve(costs)); print(f"solve (D2H + scipy)    {t:7.3f} ms")
crit._gt = gt
t, _ = T(lambda: [crit.loss_labels(o, targets, i, 10.0) for o, i in zip(outs, idx)]); print(f"loss_labels x10        {t:7.3f} ms")
t, _ = T(lambda: [crit.loss_masks(o, targets, i, 10.0, "l") for o, i in zip(outs, idx)]); print(f"loss_masks x10 (main)  {t:7.3f} ms")
from mp_former_amd.point_sample import uncertain_point_coords, map_rows, MaskLossSums
o = outs[0]; i = idx[0]
b_idx, q_idx = crit._get_src_permutation_idx(i, dev)
rows = map_rows(o["pred_masks"], (b_idx, q_idx))
t, coords = T(lambda: uncertain_point_coords(o["pred_masks"], rows, 12544, 3.0, 0.75, "x")); print(f"  uncertain coords x1  {t:7.3f} ms (n={rows.numel()})")
t, _ = T(lambda: torch.rand(rows.numel(), 37632, 2, device=dev)); print(f"    rand               {t:7.3f} ms")
lg = torch.randn(rows.numel(), 37632, device=dev)
t, _ = T(lambda: torch.topk(-lg.abs(), k=9408, dim=1)); print(f"    topk               {t:7.3f} ms")
gt_rows = torch.zeros(rows.numel(), dtype=torch.int32, device=dev)
t, _ = T(lambda: MaskLossSums.apply(o["pred_masks"], rows, gt.u8, gt_rows, coords)); print(f"  MaskLossSums fwd x1  {t:7.3f} ms")
t, _ = T(lambda: crit(out, targets)); print(f"criterion total        {t:7.3f} ms")
