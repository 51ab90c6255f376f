"""Host cost of the per-call idioms of the ctypes wrappers (device context manager, stream lookup) on the GPU box."""
import timeit, torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
n = 20000
t1 = timeit.timeit(lambda: torch.cuda.current_stream(dev).cuda_stream, number=n) / n * 1e6
t2 = timeit.timeit(lambda: torch._C._cuda_getCurrentRawStream(0), number=n) / n * 1e6
def ctx():
    with torch.cuda.device(dev):
        pass
t3 = timeit.timeit(ctx, number=n) / n * 1e6
t4 = timeit.timeit(lambda: torch.cuda.current_device() == 0, number=n) / n * 1e6
t5 = timeit.timeit(lambda: torch.empty((1024, 256), dtype=torch.float32, device=dev), number=n) / n * 1e6
print(f"current_stream().cuda_stream {t1:.2f} us | _cuda_getCurrentRawStream {t2:.2f} us | with cuda.device {t3:.2f} us | current_device()== {t4:.2f} us | torch.empty {t5:.2f} us")
