"""Developer probe: time srk_linear_bf16 (plain bf16 epilogue) for a sweep of shapes to separate read / write cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpu_superresolution_amd import ops
M = 131072
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for K, N in ((192, 192), (576, 192), (192, 576), (384, 192), (192, 384), (64, 64), (192, 64), (64, 192)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
    us = t(lambda: ops.linear_bf16(a, w, b))
    rd, wr = M * K * 2 / 1e6, M * N * 2 / 1e6
    print(f"K={K:4d} N={N:4d}: {us:7.1f} us  read {rd:6.1f} MB write {wr:6.1f} MB -> {(rd + wr) / us / 1e3:5.2f} TB/s  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
# pure copy reference
x = torch.empty(M * 192, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(x)
us = t(lambda: y.copy_(x)); print(f"torch copy 50 MB->50 MB: {us:.1f} us -> {100 / us / 1e3:.2f} TB/s")
x = torch.empty(M * 192 * 8, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(x)
us = t(lambda: y.copy_(x)); print(f"torch copy 403 MB->403 MB: {us:.1f} us -> {805 / us / 1e3:.2f} TB/s")
