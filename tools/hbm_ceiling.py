"""Developer probe: practical HBM streaming ceilings of the box (torch elementwise kernels, HIP-event timed)."""
import torch

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

for mb in (100, 400, 1600):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device="cuda"); b = torch.rand(n, device="cuda"); c = torch.empty(n, device="cuda")
    t = timeit(lambda: c.copy_(a));            print(f"{mb:5d} MB copy   (1R:1W): {t:8.1f} us  {2 * mb * 1.048576 / t:6.2f} TB/s", flush=True)
    t = timeit(lambda: torch.add(a, b, out=c)); print(f"{mb:5d} MB add    (2R:1W): {t:8.1f} us  {3 * mb * 1.048576 / t:6.2f} TB/s", flush=True)
    t = timeit(lambda: a.add_(b));             print(f"{mb:5d} MB add_   (2R:1W in place): {t:8.1f} us  {3 * mb * 1.048576 / t:6.2f} TB/s", flush=True)
    t = timeit(lambda: c.fill_(1.0));          print(f"{mb:5d} MB fill   (0R:1W): {t:8.1f} us  {mb * 1.048576 / t:6.2f} TB/s", flush=True)
    t = timeit(lambda: a.sum());               print(f"{mb:5d} MB sum    (1R:0W): {t:8.1f} us  {mb * 1.048576 / t:6.2f} TB/s", flush=True)
    del a, b, c

# pure-read ceiling with a hand-written streaming reduction (libsrk srk_grad_sumsq: float4 loads, grid-stride)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpu_superresolution_amd._lib import check, lib
for mb in (100, 400, 1600):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device="cuda"); acc = torch.zeros(1, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: check(lib().srk_grad_sumsq(a.data_ptr(), n, acc.data_ptr(), st)))
    print(f"{mb:5d} MB srk sumsq (1R:0W): {t:8.1f} us  {mb * 1.048576 / t:6.2f} TB/s", flush=True)
    del a
