import torch
dev='cuda:0'
def t(fn,n=30):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for mb in (44,176,704):
    n=mb*1024*1024//4
    xs=[torch.empty(n,device=dev) for _ in range(4)]
    ys=[torch.randn(n,device=dev) for _ in range(4)]
    i=[0]
    def fill():
        i[0]=(i[0]+1)%4; xs[i[0]].fill_(1.0)
    def copy():
        i[0]=(i[0]+1)%4; xs[i[0]].copy_(ys[i[0]])
    def rd():
        i[0]=(i[0]+1)%4; return ys[i[0]].sum()
    tf,tc,tr=t(fill),t(copy),t(rd)
    print(f"{mb} MB: fill {tf:.1f} us {mb*1.048576/tf*1e3:.0f} GB/s | copy {tc:.1f} us {2*mb*1.048576/tc*1e3:.0f} GB/s | sum {tr:.1f} us {mb*1.048576/tr*1e3:.0f} GB/s")
