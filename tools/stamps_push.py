#!/usr/bin/env python3
"""s_memtime phase stamps of the MSDA push kernel (benchmark only): mean cycles between stamps."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_msda import problem
from mp_former_amd import _lib, ms_deform_attn_backward, msda
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "init"
ab = int(sys.argv[2]) if len(sys.argv) > 2 else 0
value, shapes, lsi, loc, attn, go, S = problem("B", 2, dev, mode)
ss = msda.attach_host_shapes(shapes, shapes.tolist(), lsi)
nblocks = 2 * 8 * (16 + 64 + 256)
buf = torch.zeros(nblocks * 16, dtype=torch.int64, device=dev)
lib = _lib.lib()
lib.mpf_debug_set_buffer.argtypes = [ctypes.c_void_p]
_lib.set_option("msda_push_ablate2", ab)
for _ in range(3):
    ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
torch.cuda.synchronize()
lib.mpf_debug_set_buffer(buf.data_ptr())
ms_deform_attn_backward(value, ss, lsi, loc, attn, go, 128)
torch.cuda.synchronize()
lib.mpf_debug_set_buffer(None)
t = buf.cpu().numpy().reshape(nblocks, 16).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 14] > 0)]
d = np.diff(t[:, :15], axis=1)
names = ["prologue+issue loads", "barrier", "decode+bbox (wait loads)", "fetch0 issue", "hash count", "barrier+reserve",
         "commit0+barrier", "L0 reduce", "barrier+commit1", "L1 reduce", "barrier+commit2", "L2 reduce", "barrier", "base+barrier"]
print(f"mode={mode} ablate={ab} blocks={len(t)}  total mean {np.mean(t[:,14]-t[:,0]):.0f} cycles (100 MHz ticks? see scale)")
for n, m, md in zip(names, d.mean(0), np.median(d, 0)):
    print(f"  {n:28s} mean {m:9.0f}  median {md:9.0f}")
span = t[:, 14].max() - t[:, 0].min()
print("kernel span (ticks):", span)
