#!/usr/bin/env python3
"""Run-to-run reproducibility of the AMP head forward (tests' head_ragged fixture, draws replayed, assignment pinned): the same
forward several times in one process, outputs of the pixel decoder / the decoder / the losses compared bit for bit."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa: E402

from conftest import fifo_to_tags, load_head_fixture  # noqa: E402
from test_head_gpu import _build  # noqa: E402
from mp_former_amd import _rng  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "head_ragged"
dev = torch.device("cuda:0")
z, cfg, pp, dp, feats, targets, replay = load_head_fixture(name)
h = _build(cfg, pp, dp, dev)
feats = {k: v.to(dev) for k, v in feats.items()}
targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
matcher = h.criterion.matcher
solve = matcher.match_many
pin = []
os.environ["MPF_DEVICE_LSA"] = "0"
cap = {}


def hook_pd(mod, inp, out):
    cap["mask_features"] = out[0].detach().float().clone()
    for i, t in enumerate(out[2]):
        cap[f"ms{i}"] = t.detach().float().clone()


def run(first, backward):
    tags = fifo_to_tags(replay, cfg, True)
    if first:
        matcher.match_many = lambda *a, **k: pin.append(solve(*a, **k)) or pin[-1]
    else:
        matcher.match_many = lambda *a, **k: pin[0]
        tags = {t: d for t, d in tags.items() if not t.startswith("match")}
    _rng.install_replay(tags)
    cap.clear()
    h.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pdo = h.pixel_decoder.forward_features(feats)
        cap["mask_features"] = pdo[0].detach().float().clone()
        for i, t in enumerate(pdo[2]):
            cap[f"ms{i}"] = t.detach().float().clone()
        losses, _ = h(feats, targets)
        total = sum(losses.values())
    if backward:
        total.backward()
    _rng.install_replay(None)
    torch.cuda.synchronize()
    out = dict(cap)
    out.update({"loss." + k: v.detach().float().clone() for k, v in losses.items()})
    return out


ref = run(True, True)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    # junk allocations in between so that buffers come back with different contents
    junk = [torch.full((1 << 20,), float(it + 1) * 1e3, device=dev) for _ in range(16)]
    del junk
    cur = run(False, True)
    diff = [k for k in ref if not torch.equal(ref[k], cur[k])]
    worst = max([float((ref[k] - cur[k]).abs().max() / (ref[k].abs().max() + 1e-30)) for k in diff], default=0.0)
    print(f"run {it}: {len(diff)} of {len(ref)} tensors differ; first: {diff[:4]}; worst relative {worst:.2e}", flush=True)
