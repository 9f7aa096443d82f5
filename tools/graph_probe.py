"""Developer probe: cfg3 train step eager vs replayed as one hipGraph (timing only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpu_superresolution_amd as T  # noqa: E402
from tpu_superresolution_amd.optim import FusedAdamW  # noqa: E402
from tpu_superresolution_amd.training import train_step  # noqa: E402

torch.manual_seed(0)
model = T.SwinIR(upscale=4, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=[6] * 6, embed_dim=180, num_heads=[6] * 6,
                 mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv", drop_path_rate=0.1).cuda().train()
opt = FusedAdamW(model, lr=2e-5, weight_decay=0.0, max_grad_norm=1.0)
x = torch.rand(32, 3, 64, 64, device="cuda")
t = torch.rand(32, 3, 256, 256, device="cuda")


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    train_step(model, opt, x, t)
print(f"eager  {timed(lambda: train_step(model, opt, x, t)):.2f} ms/step", flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        train_step(model, opt, x, t)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss, bad = train_step(model, opt, x, t)
print("captured", flush=True)
print(f"graph  {timed(g.replay):.2f} ms/step  loss {float(loss):.5f} bad {int(bad)}", flush=True)
