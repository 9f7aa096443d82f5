"""Developer probe: phase timestamps of qkv_attn_fwd_kernel's workgroup 0 (library built with -DSRK_PROBE_ATTN).

    SRK_LIB_PATH=.../_variants/probe.so python tools/attn_probe.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tpu_superresolution_amd as T  # noqa: E402
from tpu_superresolution_amd import _lib  # noqa: E402
from oracle import swinir_oracle as O  # noqa: E402

cfg = O.SwinIRConfig.classical_x4()
m = T.SwinIR(drop_path_rate=0.0, **cfg.kwargs()).cuda().train()
x = torch.rand(32, 3, 64, 64, device="cuda")
for _ in range(2):
    y = m(x)
    y.mean().backward()
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(8 * 16 * 8, dtype=np.uint64)
L.srk_debug_attn_probe.argtypes = [C.c_void_p]
rc = L.srk_debug_attn_probe(buf.ctypes.data)
assert rc == 0, rc
b = buf.reshape(8, 16, 8).astype(np.int64)
t0 = b[:, 0, 0].min()
names = ["top", "Ba", "proj", "Bb", "qkvst", "attn", "Bc", "aost"]
for w in (0, 3, 4, 7):
    print(f"wave {w}: per-window phase durations in us (10 ns ticks); columns: " + " ".join(names[1:]) + " | window total")
    for t in range(8):
        r = b[w, t]
        d = (r[1:] - r[:-1]) / 100.0
        nxt = b[w, t + 1, 0] if t + 1 < 8 and b[w, t + 1, 0] else r[7]
        print(f"  t={t} start {(r[0] - t0) / 100.0:7.2f}  " + " ".join(f"{v:6.2f}" for v in d) + f" | {(nxt - r[0]) / 100.0:6.2f}")

ub = np.zeros(8 * 3 * 8, dtype=np.uint64)
L.srk_debug_attn_unit_probe.argtypes = [C.c_void_p]
assert L.srk_debug_attn_unit_probe(ub.ctypes.data) == 0
ub = ub.reshape(8, 3, 8).astype(np.int64)
print("unit marks (shader cycles) of window t=3: QK-mfma | +bias,mask | softmax | PV,scale | pack,store ; then gap to next unit")
for w in range(8):
    for k in range(3):
        r = ub[w, k]
        d = r[1:6] - r[0:5]
        gap = (ub[w, k + 1, 0] - r[5]) if k < 2 else 0
        print(f"  wave {w} unit {k}: start {r[0] - ub[:, 0, 0].min():6d}  " + " ".join(f"{v:5d}" for v in d) + f"  | total {r[5] - r[0]:5d} gap {gap}")
