#!/usr/bin/env python3
"""s_memtime phase sums of the MSDA tile kernel (benchmark only; mpf_debug_set_buffer): per wave cycles spent waiting for the
chunk's rows, in the corner dot products, loading the MFMA operands, preparing the next chunk, and in the MFMAs."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_msda import problem
from mp_former_amd import _lib, ms_deform_attn_backward, msda
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "init"
value, shapes, lsi, loc, attn, go, S = problem("B", 2, dev, mode)
ss = msda.attach_host_shapes(shapes, shapes.tolist(), lsi)
nw = 2 * 8 * 4096 * 4
buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.mpf_debug_set_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
torch.cuda.synchronize()
lib.mpf_debug_set_buffer(buf.data_ptr())
ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
torch.cuda.synchronize()
lib.mpf_debug_set_buffer(None)
t = buf.cpu().numpy().reshape(nw, 8).astype(np.int64)
t = t[t[:, 7] > 0]
ch = t[:, 7]
print(f"mode={mode}: {len(t)} waves with work, {ch.sum()} chunks ({ch.mean():.2f} per wave); ticks are s_memtime units (100 MHz)")
print(f"  prologue (start -> loop entry)     mean {np.mean(t[:, 1] - t[:, 0]):8.1f} per wave")
for k, n in enumerate(["wait rows", "corner dots", "mfma operands", "next chunk (sample role, DMA, gathers)", "32 mfma"]):
    print(f"  {n:40s} {t[:, 2 + k].sum() / ch.sum():8.1f} per chunk")
print(f"  total per chunk {t[:, 2:7].sum() / ch.sum():.1f};  kernel span {t[:, 0].max() - t[:, 0].min()} ticks (last wave start - first wave start)")
for lo, hi, nm in ((1, 3, "<=3 chunks"), (4, 6, "4-6"), (7, 99, ">=7")):
    s_ = t[(ch >= lo) & (ch <= hi)]
    if len(s_):
        print(f"  waves with {nm}: {len(s_)}, per-chunk total {s_[:, 2:7].sum() / s_[:, 7].sum():.1f}, prologue {np.mean(s_[:, 1] - s_[:, 0]):.1f}")
