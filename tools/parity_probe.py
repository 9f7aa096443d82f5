"""GPU probe (developer tool): (1) cfg3 / cfg2 forward parity vs the fp32 oracle with default-scale weights (scale=1.0:
N(0, 0.02) linears like the reference's init) against SURVEY 8c's 5e-3 max-abs / 60 dB mutual-PSNR targets; (2) whether
two identical backward passes give bit-identical gradients, and which tensors differ if not."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from oracle import swinir_oracle as O  # noqa: E402
import tpu_superresolution_amd as T  # noqa: E402


def build(cfg, sd, train=False):
    m = T.SwinIR(drop_path_rate=0.0, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    return m.train() if train else m.eval()


def main():
    torch.set_num_threads(16)
    for tag, cfg, hw, bs in (("cfg3", O.SwinIRConfig.classical_x4(), 64, 2), ("cfg2", O.SwinIRConfig.light_x2(), 48, 4)):
        for scale in (1.0, 1.5, 3.0):
            sd = O.random_state_dict(cfg, seed=42, scale=scale)
            x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(0))
            with torch.no_grad():
                y = build(cfg, sd)(x.cuda()).cpu()
                ref = O.swinir_forward(sd, cfg, x)
            err = float((y - ref).abs().max())
            mse = float(((y - ref) ** 2).mean())
            print(f"[parity] {tag} weight scale {scale}: max|ref| {float(ref.abs().max()):.3f} std {float(ref.std()):.3f} "
                  f"max abs err {err:.3e} mutual PSNR {10 * np.log10(1.0 / mse):.1f} dB", flush=True)
    # determinism of gradients
    cfg = O.SwinIRConfig.classical_x4()
    sd = O.random_state_dict(cfg, seed=42, scale=1.5)
    x = torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(3)).cuda()
    t = torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(4)).cuda()
    m = build(cfg, sd, train=True)
    runs = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.l1_loss(m(x), t).backward()
        runs.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
    diff = [n for n in runs[0] if not torch.equal(runs[0][n], runs[1][n])]
    kinds = sorted({".".join(n.split(".")[-2:]) if "blocks" in n else n for n in diff})
    print(f"[determinism] {len(diff)} of {len(runs[0])} gradient tensors differ between two identical backward passes: {kinds}")


if __name__ == "__main__":
    main()
