"""Developer tool: the re-projecting attention backward (csrc/attn_bwd_fused.hip) alone at the cfg3 shape (2048 windows).

    python tools/abf_bench.py [reps]                       event-timed launches
    SRK_LIB_PATH=.../_variants/abf_probe.so python tools/abf_bench.py 3 --probe     phase timestamps of workgroup 0
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tpu_superresolution_amd import _lib, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
B, H, W, shift = 32, 64, 64, 4
B_ = B * (H // 8) * (W // 8)
g = torch.Generator().manual_seed(0)
xn = (torch.randn(B_ * 64, 192, generator=g)).to(torch.bfloat16).cuda()
dx = (torch.randn(B_ * 64, 192, generator=g) * 0.5).to(torch.bfloat16).cuda()
wq = (torch.randn(576, 192, generator=g) * 0.08).to(torch.bfloat16).cuda()
bq = (torch.randn(576, generator=g) * 0.2).cuda()
wp = (torch.randn(192, 192, generator=g) * 0.08).to(torch.bfloat16).cuda()
biasd = ops.rel_pos_bias_expand((torch.randn(225, 6, generator=g) * 0.5).cuda())
for _ in range(3):
    ops.window_attention_bwd_fused(xn, wq, bq, 30 ** -0.5, dx, wp, biasd, H, W, shift)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    ops.window_attention_bwd_fused(xn, wq, bq, 30 ** -0.5, dx, wp, biasd, H, W, shift)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
print(f"fused attention backward + table reduce: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us over {reps} launches "
      f"(250 MB algorithmic: {250e6 / (ts[len(ts) // 2] * 1e-6) / 1e12:.2f} TB/s)")
if "--probe" in sys.argv:
    L = _lib.lib()
    buf = np.zeros(12 * 16 * 8, dtype=np.uint64)
    L.srk_debug_abf_probe.argtypes = [C.c_void_p]
    assert L.srk_debug_abf_probe(buf.ctypes.data) == 0
    b = buf.reshape(12, 16, 8).astype(np.int64)
    t0 = b[:, 0, 0].min()
    names = ["waitA", "proj", "waitB", "phaseB", "waitC", "phaseC"]
    for w in (0, 1, 3, 5, 11):
        print(f"wave {w}: per-window durations in us; " + " ".join(names) + " | window")
        for t in range(16):
            r = b[w, t]
            d = (r[1:7] - r[0:6]) / 100.0
            nxt = b[w, t + 1, 0] if t + 1 < 16 and b[w, t + 1, 0] else r[6]
            print(f"  t={t:2d} start {(r[0] - t0) / 100.0:7.2f}  " + " ".join(f"{v:6.2f}" for v in d) + f" | {(nxt - r[0]) / 100.0:6.2f}")
