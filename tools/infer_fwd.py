"""Developer probe: cfg3-size forward passes in eval mode (no q/k/v stores) for a kernel trace."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpu_superresolution_amd as T  # noqa: E402
from oracle import swinir_oracle as O  # noqa: E402

cfg = O.SwinIRConfig.classical_x4()
m = T.SwinIR(drop_path_rate=0.0, **cfg.kwargs()).cuda().eval()
x = torch.rand(32, 3, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(6):
        y = m(x)
torch.cuda.synchronize()
print("ok", float(y.mean()))
