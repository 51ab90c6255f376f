#!/usr/bin/env python3
"""The four plain weight gradients of an encoder layer as one grouped launch (encoder_fused._wgrad_group) at R rows:
256 x 256 tiles (csrc/gemm3_nt2.h) against the 128 x 128 tiles (`gemm3_nt2=0`).   python tools/bench_nt_group.py [R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mp_former_amd import _lib  # noqa: E402
from mp_former_amd.encoder_fused import _wgrad_group  # noqa: E402
from mp_former_amd.gemm3 import amax  # noqa: E402

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 43008
torch.manual_seed(0)
NB = 3
sets = []
for _ in range(NB):
    ds2, h, dh, x1, ds1, ao, gv, x = (torch.randn(R, c, device=dev) for c in (256, 1024, 1024, 256, 256, 256, 256, 256))
    pairs = [(ds2, h), (dh, x1), (ds1, ao), (gv, x)]
    sets.append((pairs, [(amax(a), amax(b)) for a, b in pairs]))
ref = None
for on in (0, 1):
    _lib.set_option("gemm3_nt2", on)
    it = [0]

    def run():
        it[0] = (it[0] + 1) % NB
        return _wgrad_group(*sets[it[0]])
    for _ in range(3):
        run()
    _lib.profile_enable(True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    n, ms, by = _lib.profile_get("gemm3_nt_group_kernel")
    nr, msr, _ = _lib.profile_get("nt_reduce")
    _lib.profile_enable(False)
    print(f"gemm3_nt2={on}: group kernel {ms / max(n, 1) * 1e3:7.1f} us ({n} launches)  reduce {msr / max(nr, 1) * 1e3:6.1f} us  wall per call {e0.elapsed_time(e1) / 12 * 1e3:7.1f} us", flush=True)
    it[0] = 0
    res = _wgrad_group(*sets[1])
    if ref is None:
        ref = res
    else:
        for (a, b), (c, d) in zip(ref, res):
            print("   max rel diff dW", float((a - c).abs().max() / a.abs().max()), " db", float((b - d).abs().max() / b.abs().max()))
_lib.set_option("gemm3_nt2", 1)
