import sys, torch
sys.path.insert(0, "/root/repo")
import bench
import torch.nn.functional as F
dev = torch.device("cuda", 0)
m = bench.build_infer_model("cfg5", dev).train()
for bs in (2, 16):
    x, t = bench.synthetic_batch(bs, dev, seed=1000)
    m.zero_grad(set_to_none=True)
    y = m(x)
    print("bs", bs, "y finite", bool(torch.isfinite(y).all()), float(y.abs().max()))
    loss = F.l1_loss(y, t)
    loss.backward()
    bad = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in m.named_parameters() if not torch.isfinite(p.grad).all()]
    print(" loss", float(loss), "non-finite grads:", len(bad), bad[:12])
    big = sorted(((float(p.grad.norm()), n) for n, p in m.named_parameters()), reverse=True)[:5]
    print(" largest", big)
