#!/usr/bin/env python3
"""Which stage of the head's AMP step is not bit-reproducible run to run (config B, N = 2)?  Runs every stage twice on
identical inputs / draws and compares bit for bit; then the gradients of two full steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd.head import MPFormerHead  # noqa: E402


def same(a, b):
    if isinstance(a, torch.Tensor):
        return bool(torch.equal(a, b))
    return all(same(x, y) for x, y in zip(a, b))


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    H = W = 1024
    n = 2
    shapes = {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}
    h = MPFormerHead(num_classes=80, num_queries=100, feature_shapes=shapes).to(dev).train()
    feats = {k: torch.randn(n, H // s, W // s, c, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2) for k, (c, s) in shapes.items()}
    targets = []
    for b in range(n):
        T = 5 + 9 * b
        m = torch.zeros(T, H, W, dtype=torch.bool, device=dev)
        for t in range(T):
            m[t, 60 * t:60 * t + 180, 50 * t:50 * t + 260] = True
        targets.append({"labels": (torch.arange(T, device=dev) * 7) % 80, "masks": m, "boxes": torch.zeros(T, 4, device=dev)})
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with torch.autocast("cuda", dtype=torch.bfloat16):
        outs = [h.pixel_decoder.forward_features(feats) for _ in range(rep)]
        print("pixel decoder forward bit-equal:", [same((o[0], o[2]), (outs[0][0], outs[0][2])) for o in outs[1:]])
        mf, _, ms = outs[0]
        dn = {"tgt": targets, "scalar": 1, "noise_scale": 0.0}
        res = []
        for _ in range(rep):
            torch.manual_seed(11)
            o = h.predictor(ms, mf, None, dn)
            res.append(o)
        key = lambda o: [o["pred_logits"], o["pred_masks"].me] + [a["pred_logits"] for a in o["aux_outputs"]]  # noqa: E731
        print("decoder forward bit-equal:", [same(key(o), key(res[0])) for o in res[1:]])
        ls = []
        for _ in range(rep):
            torch.manual_seed(12)
            ls.append(h.criterion(res[0], targets))
        print("criterion forward bit-equal:", [all(float(l[k]) == float(ls[0][k]) for k in ls[0]) for l in ls[1:]])
    grads = []
    for _ in range(rep):
        h.zero_grad(set_to_none=True)
        torch.manual_seed(13)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = h.total_loss(feats, targets)
        loss.backward()
        torch.cuda.synchronize()
        grads.append((float(loss), {k: p.grad.clone() for k, p in h.named_parameters()}))
    print("total loss:", [g[0] for g in grads])
    for i in range(1, rep):
        bad = [k for k in grads[0][1] if not torch.equal(grads[0][1][k], grads[i][1][k])]
        groups = {}
        for k in bad:
            g = ".".join(k.split(".")[:2])
            groups[g] = groups.get(g, 0) + 1
        print(f"run {i}: {len(bad)} of {len(grads[0][1])} parameter gradients differ:", groups)


if __name__ == "__main__":
    main()
