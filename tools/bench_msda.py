#!/usr/bin/env python3
"""Micro-benchmark of the MSDA kernels at BASELINE config B (1024x1024: S=Lq=21504, M=8, D=32,
L=3, P=4).  Times each variant with HIP events on the launch stream, interleaved rounds, and prints
achieved ALGORITHMIC GB/s (fwd 800*e*S, bwd 1344*e*S bytes per image — SURVEY.md §8(d))."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib, ms_deform_attn_backward, ms_deform_attn_forward  # noqa: E402

LEVELS = {"A": [(8, 8), (16, 16), (32, 32)], "B": [(32, 32), (64, 64), (128, 128)],
          "D": [(20, 20), (40, 40), (80, 80)], "E": [(32, 64), (64, 128), (128, 256)]}


def problem(cfg, N, dev, mode):
    lv = LEVELS[cfg]
    M, D, L, P = 8, 32, 3, 4
    shapes = torch.tensor(lv, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(0)
    value = torch.randn(N, S, M, D, device=dev)
    # reference points = pixel centres of each query's own level (PIX:141-153)
    refs = []
    for (h, w) in lv:
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h), torch.linspace(0.5, w - 0.5, w), indexing="ij")
        refs.append(torch.stack((xs.reshape(-1) / w, ys.reshape(-1) / h), -1))
    ref = torch.cat(refs, 0).to(dev)  # [S,2]
    if mode == "init":
        # offsets of a freshly initialised MSDeformAttn: (cos,sin)(2*pi*m/8)/max * (p+1) pixels
        import math
        th = torch.arange(M, dtype=torch.float32) * (2 * math.pi / M)
        g = torch.stack([th.cos(), th.sin()], -1)
        g = g / g.abs().max(-1, keepdim=True)[0]
        off = g.view(M, 1, 1, 2).repeat(1, L, P, 1) * torch.arange(1, P + 1).view(1, 1, P, 1)
        off = off.to(dev)[None, None].expand(N, S, M, L, P, 2)
    elif mode == "trained":
        # offsets spread like a trained model: N(0, 3 px) around the init pattern
        off = torch.randn(N, S, M, L, P, 2, device=dev) * 3.0
    else:  # uniform random locations over the whole map (worst-case locality)
        off = None
    if off is None:
        loc = torch.rand(N, S, M, L, P, 2, device=dev)
    else:
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
        loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    attn = torch.softmax(torch.randn(N, S, M, L * P, device=dev), -1).view(N, S, M, L, P)
    go = torch.randn(N, S, M * D, device=dev)
    return value, shapes, lsi, loc.contiguous(), attn, go, S


def time_fn(fn, iters):
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="B")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--mode", default="init", choices=["init", "trained", "uniform"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--routes", default="host,dev,atomic",
                    help="host = blocked kernels, geometry from attached host shapes (the module's own route); dev = the reference's "
                         "all-device B1 signature, geometry built by a prologue kernel (mpf_msda_*_dev); atomic = round-1 kernels")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    value, shapes, lsi, loc, attn, go, S = problem(a.cfg, a.batch, dev, a.mode)
    N = a.batch
    fwd_bytes = 800 * 4 * S * N
    bwd_bytes = 1344 * 4 * S * N
    res = {}
    from mp_former_amd import msda as msda_mod
    shapes_host = msda_mod.attach_host_shapes(shapes.clone(), LEVELS[a.cfg], lsi)
    for rnd in range(a.rounds + 1):
        for route in a.routes.split(","):
            msda_mod.BWD_MODE = "atomic" if route == "atomic" else "auto"
            sh = shapes_host if route == "host" else shapes
            f = lambda: ms_deform_attn_forward(value, sh, lsi, loc, attn, 128)  # noqa: E731
            b = lambda: ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 128)  # noqa: E731
            if route == "atomic":
                _lib.set_option("msda_block_disable", 1)
            f(); b(); torch.cuda.synchronize()
            tf = time_fn(f, a.iters)
            kf = _lib.last_kernel()
            tb = time_fn(b, a.iters)
            kb = _lib.last_kernel()
            _lib.set_option("msda_block_disable", 0)
            if rnd > 0:
                res.setdefault((route, "fwd", kf), []).append(tf)
                res.setdefault((route, "bwd", kb), []).append(tb)
    msda_mod.BWD_MODE = "auto"
    print(f"cfg={a.cfg} N={N} S={S} mode={a.mode}  fwd_alg={fwd_bytes/1e6:.1f} MB  bwd_alg={bwd_bytes/1e6:.1f} MB")
    med = {}
    for (route, d, k), ts in res.items():
        ts = sorted(ts)
        med[(route, d)] = ts[len(ts) // 2]
        by = fwd_bytes if d == "fwd" else bwd_bytes
        print(json.dumps({"route": route, "dir": d, "kernel": k, "us_med": round(med[(route, d)], 1), "us_min": round(ts[0], 1),
                          "alg_GBps": round(by / med[(route, d)] / 1e3, 1), "frac_of_8TBps": round(by / med[(route, d)] / 1e3 / 8000, 4)}))
    if ("host", "fwd") in med and ("dev", "fwd") in med:
        print(json.dumps({"dev_over_host": {"fwd": round(med[("dev", "fwd")] / med[("host", "fwd")], 3),
                                            "bwd": round(med[("dev", "bwd")] / med[("host", "bwd")], 3)}}))


if __name__ == "__main__":
    main()
