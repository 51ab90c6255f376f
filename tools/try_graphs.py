#!/usr/bin/env python3
"""Which piece of the static part of the step captures as a HIP graph (one piece per process: a failed capture may crash).
usage: try_graphs.py A|B|C|conv|fold"""
import os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mp_former_amd import _lib, _miopen
from mp_former_amd import graphs as G
dev = torch.device("cuda:0")
_lib.lib(); _miopen.use_shipped_find_db(check_version=False)
what = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
model = bench.TrainModel().to(dev).train()
model.backbone.to(memory_format=torch.channels_last)
img, tg = bench.synth_batch(2, size, 80, 1, dev)
x = img.contiguous(memory_format=torch.channels_last)
mk = torch.cuda.make_graphed_callables
import traceback
for cls in (G._StagesA, G._StagesB, G._PixelDecoder):
    def wrap(cls):
        orig = cls.forward
        def fwd(self, *a):
            try:
                return orig(self, *a)
            except BaseException:
                traceback.print_exc()
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(3)
        cls.forward = fwd
    wrap(cls)
_bw = torch.autograd.grad
def _grad(*a, **k):
    try:
        return _bw(*a, **k)
    except BaseException:
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(4)
torch.autograd.grad = _grad
with torch.autocast("cuda", dtype=torch.bfloat16):
    feats = model.backbone(x)
f = {k: v.detach().clone().requires_grad_(True) for k, v in feats.items()}
if what == "A":
    g = mk(G._StagesA(model.backbone), (x,), allow_unused_input=True)
    o = g(x)
elif what == "B":
    g = mk(G._StagesB(model.backbone), (f["res3"],), allow_unused_input=True)
    o = g(f["res3"])
elif what == "C":
    g = mk(G._PixelDecoder(model.head.pixel_decoder), (f["res2"], f["res3"], f["res4"], f["res5"]), allow_unused_input=True)
    o = g(f["res2"], f["res3"], f["res4"], f["res5"])
elif what == "conv":          # one MIOpen convolution alone
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    xi = torch.randn(2, 64, 64, 64, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    class M(torch.nn.Module):
        def __init__(s): super().__init__(); s.c = conv
        def forward(s, t):
            with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
                return s.c(t)
    g = mk(M(), (xi,), allow_unused_input=True)
    o = (g(xi),)
elif what == "fold":          # the grouped fold-cast + bias_act kernels without any convolution
    bb = model.backbone
    class M(torch.nn.Module):
        def __init__(s): super().__init__(); s.bb = bb
        def forward(s, t):
            with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
                fo = s.bb._fold_group([p for b in s.bb.res2 for p in b.pairs()], torch.bfloat16)
                return t * fo[0][0].float().sum() + fo[1][0].float().sum()
    xi = torch.randn(8, device=dev, requires_grad=True)
    g = mk(M(), (xi,), allow_unused_input=True)
    o = (g(xi),)
sum(t.float().sum() for t in o).backward()
torch.cuda.synchronize()
print("captured + replayed:", what)
