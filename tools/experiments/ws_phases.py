#!/usr/bin/env python3
"""Where a tile of the weight-stationary GEMM kernel spends its time (s_memtime stamps of wave 0 of every workgroup).
Builds a -DG3_TIMING copy of the library into /tmp first.   python tools/ws_phases.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "mp_former_amd", "csrc")
if "MPF_LIB_PATH" not in os.environ:
    objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f != "gemm3.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-DG3_TIMING", "-DWS_TIME_WAVE=" + (sys.argv[1] if len(sys.argv) > 1 else "0"), *sys.argv[2:],
                           "-c", os.path.join(CS, "gemm3.hip"), "-o", "/tmp/gemm3_timing.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", "/tmp/libmpf_timing.so"] + objs + ["/tmp/gemm3_timing.o"])
    os.environ["MPF_LIB_PATH"] = "/tmp/libmpf_timing.so"
    os.execv(sys.executable, [sys.executable] + sys.argv)      # (nothing has touched the GPU yet)
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mp_former_amd import _lib  # noqa: E402
from mp_former_amd.gemm3 import amax, gemm3_h2, split_weights_grouped_h2  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib()
fn = lib.mpf_gemm3_debug_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
M = 43008
_lib.set_option("gemm3_ws", 256)
for N, cinf in ((256, False), (256, True), (1024, False)):
    a = torch.randn(M, 256, device=dev)
    w = torch.randn(N, 256, device=dev) / 16
    cin = torch.randn(M, N, device=dev) if cinf else None
    (pl, wam), = split_weights_grouped_h2([([w], False)])
    am = amax(a)
    gemm3_h2(a, am, pl, wam, cin=cin)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    fn(buf, 1)
    for _ in range(10):
        gemm3_h2(a, am, pl, wam, cin=cin)
    torch.cuda.synchronize()
    fn(buf, 1)
    tiles = buf[7]
    names = ["prologue(per WG)", "epi loads", "compute", "wait rows", "convert", "epilogue", "dma+barrier"]
    wgs = tiles / (10 * 5.5) if N == 256 else tiles / (10 * 21)
    print(f"N={N} cin={cinf}: ticks per tile of wave 0: " + ", ".join(f"{nm} {buf[i] / (tiles if i else wgs * 10):.0f}" for i, nm in enumerate(names)), flush=True)
