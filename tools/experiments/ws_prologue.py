import ctypes, os, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
CS = os.path.join(ROOT, "mp_former_amd", "csrc")
if "MPF_LIB_PATH" not in os.environ:
    objs = [os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".o") and f != "gemm3.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-DG3_TIMING", "-DWS_TIME_PROLOGUE=1", *sys.argv[1:], "-c", os.path.join(CS, "gemm3.hip"), "-o", "/tmp/gemm3_timing.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", "/tmp/libmpf_timing.so"] + objs + ["/tmp/gemm3_timing.o"])
    os.environ["MPF_LIB_PATH"] = "/tmp/libmpf_timing.so"
    os.execv(sys.executable, [sys.executable] + sys.argv)
sys.path.insert(0, ROOT)
import torch
from mp_former_amd import _lib
from mp_former_amd.gemm3 import amax, gemm3_h2, split_weights_grouped_h2
dev = torch.device("cuda:0")
lib = _lib.lib()
fn = lib.mpf_gemm3_debug_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
_lib.set_option("gemm3_ws", 256)
M = 43008
for N in (256, 1024):
    a = torch.randn(M, 256, device=dev); w = torch.randn(N, 256, device=dev) / 16
    (pl, wam), = split_weights_grouped_h2([([w], False)])
    am = amax(a)
    gemm3_h2(a, am, pl, wam); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)(); fn(buf, 1)
    for _ in range(10): gemm3_h2(a, am, pl, wam)
    torch.cuda.synchronize(); fn(buf, 1)
    n = buf[7]
    print(f"N={N}: prologue cycles per WG (wave 0): issue {buf[1]/n:.0f}, wait for all {buf[2]/n:.0f}, convert+dma+barrier {buf[3]/n:.0f}")
