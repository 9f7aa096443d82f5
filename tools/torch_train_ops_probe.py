"""Developer probe: which torch ops (not libsrk launches) a HAT / DAT TRAIN step issues, by Python call site.

    python tools/torch_train_ops_probe.py cfg5

A TorchDispatchMode counts every aten op of one eager train step against the innermost frame inside the package (forward and the
manual backward; autograd-engine internals show up under the frame that called .backward / autograd.grad).
"""
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tpu_superresolution_amd.training import l1_loss_checked  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda")
m = bench.build_infer_model(name, dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=2e-5, weight_decay=0.0)
lr_img, hr_img = bench.synthetic_batch(16, dev, seed=1000)


def step():
    opt.zero_grad(set_to_none=True)
    loss, bad = l1_loss_checked(m(lr_img), hr_img)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_op = collections.Counter()
        self.by_site = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        op = str(func).replace("aten.", "")
        f = sys._getframe(0)
        site = "?"
        while f is not None:
            fn = f.f_code.co_filename
            if "tpu_superresolution_amd" in fn and "_python_dispatch" not in fn:
                site = f"{os.path.basename(fn)}:{f.f_lineno} {f.f_code.co_name}"
                break
            f = f.f_back
        self.by_op[op] += 1
        self.by_site[site][op] += 1
        return func(*args, **(kwargs or {}))


for _ in range(2):
    step()
torch.cuda.synchronize()
c = Count()
with c:
    step()
torch.cuda.synchronize()
print("aten ops of one step:", sum(c.by_op.values()))
for k, v in c.by_op.most_common(40):
    print(f"  {v:6d}  {k}")
print("by call site:")
for site, ops in sorted(c.by_site.items(), key=lambda kv: -sum(kv[1].values()))[:70]:
    tot = sum(ops.values())
    print(f"  {tot:6d}  {site}: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(6)))
