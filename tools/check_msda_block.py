#!/usr/bin/env python3
"""Development check of the spatially blocked MSDA kernels (csrc/msda_block.hip): parity against the C
oracle on several problem classes (pixel-decoder-like offsets, uniform random, out-of-range, odd level
sizes, queries that are not the pixels, run overflow), then timings at config B / E.
Test infrastructure: imports oracle/ as the checker only."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mp_former_amd import _lib, msda  # noqa: E402
from tools.bench_msda import LEVELS, problem, time_fn  # noqa: E402


def make(lv, N, Lq=None, mode="uniform", seed=0, M=8, D=32, P=4, spread=3.0):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.tensor(lv, dtype=torch.long)
    L = len(lv)
    S = int(shapes.prod(1).sum())
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, generator=g)
    Lq = S if Lq is None else Lq
    if mode == "uniform":
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15
    elif mode == "point":
        loc = torch.full((N, Lq, M, L, P, 2), 0.37) + torch.rand(N, Lq, M, L, P, 2, generator=g) * 0.01
    else:
        assert Lq == S
        refs = []
        for (h, w) in lv:
            ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
            refs.append(torch.stack((xs.reshape(-1) / w, ys.reshape(-1) / h), -1))
        ref = torch.cat(refs, 0)
        off = torch.randn(N, S, M, L, P, 2, generator=g) * spread
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
        loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g)
    return dict(value=value, shapes=shapes, lsi=lsi, loc=loc.contiguous(), attn=attn, go=go)


def run_gpu(z, dev):
    ss = msda.attach_host_shapes(z["shapes"].to(dev), z["shapes"].tolist(), z["lsi"].to(dev))
    lsi = ss._mpf_lsi
    v, loc, a, go = (z[k].to(dev) for k in ("value", "loc", "attn", "go"))
    out = msda.ms_deform_attn_forward(v, ss, lsi, loc, a, 128)
    kf = _lib.last_kernel()
    gv, gl, ga = msda.ms_deform_attn_backward(v, ss, lsi, loc, a, go, 128)
    kb = _lib.last_kernel()
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (out, gv, gl, ga)], kf, kb


def smooth(z, eps=1e-4):
    loc = z["loc"].double().numpy()
    sh = z["shapes"].double().numpy()
    x = loc[..., 0] * sh[None, None, None, :, None, 1] - 0.5
    y = loc[..., 1] * sh[None, None, None, :, None, 0] - 0.5
    return ~((np.abs(x - np.round(x)) < eps) | (np.abs(y - np.round(y)) < eps))


def err(a, b):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return float(d.max()), float(d.max() / (np.abs(b).max() + 1e-30))


def parity(dev):
    from oracle import msda_oracle as O
    cases = [
        ("A-like near", [(8, 8), (16, 16), (32, 32)], 2, None, "near", 3.0),
        ("A uniform+oob", [(8, 8), (16, 16), (32, 32)], 2, None, "uniform", 0),
        ("D-like odd 20/40/80 near", [(20, 20), (40, 40), (80, 80)], 1, None, "near", 2.0),
        ("odd 6x4,3x2,12x9 uniform", [(6, 4), (3, 2), (12, 9)], 2, None, "uniform", 0),
        ("1 level 13x7", [(13, 7)], 1, None, "near", 1.5),
        ("4 levels", [(4, 4), (8, 8), (16, 16), (32, 32)], 1, None, "near", 2.0),
        ("Lq != S (300 queries)", [(8, 8), (16, 16), (32, 32)], 2, 300, "uniform", 0),
        ("overflow: all samples at one point", [(8, 8), (16, 16), (32, 32)], 1, None, "point", 0),
        ("E-aspect 16x32.. near wide", [(16, 32), (32, 64), (64, 128)], 1, None, "near", 6.0),
    ]
    ok = True
    for name, lv, N, Lq, mode, spread in cases:
        z = make(lv, N, Lq, mode, seed=len(name), spread=spread)
        (out, gv, gl, ga), kf, kb = run_gpu(z, dev)
        npz = {k: v.numpy() for k, v in z.items()}
        ro = O.msda_forward(npz["value"], npz["shapes"], npz["lsi"], npz["loc"], npz["attn"])
        rgv, rgl, rga = O.msda_backward(npz["value"], npz["shapes"], npz["lsi"], npz["loc"], npz["attn"], npz["go"])
        sm = smooth(z)
        e = {"out": err(out, ro), "gv": err(gv, rgv), "ga": err(ga, rga), "gl": err(gl[sm], rgl[sm])}
        good = e["out"][1] < 2e-5 and e["gv"][1] < 1e-4 and e["ga"][1] < 1e-4 and e["gl"][1] < 2e-3
        ok &= good
        print(("PASS " if good else "FAIL ") + f"{name:40s} fwd={kf} bwd={kb} " +
              " ".join(f"{k}:{v[0]:.2e}/{v[1]:.1e}" for k, v in e.items()), flush=True)
    return ok


def timing(dev, cfg, N, mode, iters=20, rounds=5):
    value, shapes, lsi, loc, attn, go, S = problem(cfg, N, dev, mode)
    ss = msda.attach_host_shapes(shapes, shapes.tolist(), lsi)
    res = {}
    for rnd in range(rounds + 1):
        for variant in ("block", "old"):
            _lib.set_option("msda_block_disable", 0 if variant == "block" else 1)
            f = lambda: msda.ms_deform_attn_forward(value, ss, lsi, loc, attn, 128)  # noqa: E731
            b = lambda: msda.ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)  # noqa: E731
            f(); b(); torch.cuda.synchronize()
            tf = time_fn(f, iters)
            kf = _lib.last_kernel()
            tb = time_fn(b, iters)
            kb = _lib.last_kernel()
            if rnd:
                res.setdefault(kf, []).append(tf)
                res.setdefault(kb, []).append(tb)
    _lib.set_option("msda_block_disable", 0)
    fb, bb = 800 * 4 * S * N, 1344 * 4 * S * N
    for k, ts in res.items():
        med = sorted(ts)[len(ts) // 2]
        by = fb if "fwd" in k else bb
        print(json.dumps({"cfg": cfg, "N": N, "mode": mode, "kernel": k, "us": round(med, 1),
                          "frac_of_8TBps": round(by / med / 1e3 / 8000, 4)}), flush=True)
    # per-kernel split of the backward
    _lib.profile_enable(True)
    for _ in range(5):
        msda.ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
    torch.cuda.synchronize()
    for k in ("msda_bwd_bin", "msda_bwd_tile"):
        n, ms, _ = _lib.profile_get(k)
        if n:
            print(f"   {k}: {ms / n * 1e3:.1f} us")
    _lib.profile_enable(False)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--region-rows", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.region_rows:
        _lib.set_option("msda_region_rows", a.region_rows)
    good = True
    if not a.no_parity:
        good = parity(dev)
    if not a.no_timing:
        for cfg, N, mode in (("B", 2, "init"), ("B", 2, "trained"), ("B", 2, "uniform"), ("E", 2, "init")):
            timing(dev, cfg, N, mode)
    sys.exit(0 if good else 1)
