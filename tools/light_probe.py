"""Developer probe: phase timestamps of swin_block_light_kernel (library built with -DSRK_PROBE_LIGHT)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tpu_superresolution_amd import _lib  # noqa: E402

m = bench.build_infer_model("cfg2", torch.device("cuda"))
x = torch.rand(16, 3, 48, 48, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m(x)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(2 * 4 * 16, dtype=np.uint64)
L.srk_debug_light_probe.argtypes = [C.c_void_p]
assert L.srk_debug_light_probe(buf.ctypes.data) == 0
b = buf.reshape(2, 4, 16).astype(np.int64)
names = "load LN1 bar proj bar attn bar projG bar LN2 fc1 bar fc2 store".split()
for blk in range(2):
    t0 = b[blk, :, 0].min()
    for w in range(4):
        r = b[blk, w]
        print(f"blk {blk} wave {w}: start {(r[0]-t0)/100:5.2f} " + " ".join(f"{n}={(r[i+1]-r[i])/100:4.2f}" for i, n in enumerate(names)) + f" | total {(r[14]-r[0])/100:6.2f} us")
