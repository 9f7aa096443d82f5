"""Developer probe: run the HIP model against the CPU oracle on the GPU box and print error metrics.

    python tools/gpu_probe.py [tiny|cfg2|cfg3|grads|all]

Not part of the product or of the test-suite; it prints numbers used to calibrate test tolerances.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import swinir_oracle as O  # noqa: E402
import tpu_superresolution_amd as T  # noqa: E402


def build(cfg, sd, train=False, drop_path_rate=0.0):
    m = T.SwinIR(drop_path_rate=drop_path_rate, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    return m.train() if train else m.eval()


def metrics(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    mse = float(((got - ref) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    return f"max_abs={float(err.max()):.3e} mean_abs={float(err.mean()):.3e} ref_absmax={float(ref.abs().max()):.3e} mutual_psnr={psnr:.1f}dB"


def tiny():
    from test_oracle_golden import VARIANTS, tiny_weights
    for tag in ("ps4", "psd2", "ps3"):
        g, cfg, sd = tiny_weights(tag)
        m = build(cfg, sd)
        for hw in ((16, 16), (13, 19), (24, 32)):
            x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
            with torch.no_grad():
                y = m(x.cuda())
            print(f"tiny {tag} {hw}: {metrics(y, torch.from_numpy(g[f'y_{hw[0]}x{hw[1]}']))}", flush=True)


def big(tag):
    cfg, hw, ws, bs = (O.SwinIRConfig.light_x2(), 48, 1.0, 2) if tag == "cfg2" else (O.SwinIRConfig.classical_x4(), 64, 1.5, 1)
    sd = O.random_state_dict(cfg, 42, ws)
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        ref = O.swinir_forward(sd, cfg, x)
    m = build(cfg, sd)
    with torch.no_grad():
        y = m(x.cuda())
    torch.cuda.synchronize()
    print(f"{tag} bs{bs}: {metrics(y, ref)}", flush=True)
    for name in ("f0", "x0"):
        pass


def grads():
    from test_oracle_golden import tiny_weights
    for tag in ("ps4", "psd2"):
        g, cfg, sd = tiny_weights(tag)
        m = build(cfg, sd, train=True)
        x, t = torch.from_numpy(g["train.x"]).cuda(), torch.from_numpy(g["train.target"]).cuda()
        out = m(x)
        loss = (out - t).abs().mean()
        loss.backward()
        print(f"grads {tag}: loss {float(loss):.6f} ref {float(g['train.loss']):.6f}", flush=True)
        worst = []
        for n, p in m.named_parameters():
            ref = torch.from_numpy(g["grad." + n])
            got = p.grad.float().cpu()
            rel = float((got - ref).norm() / (ref.norm() + 1e-12))
            worst.append((rel, n, float(ref.norm())))
        worst.sort(reverse=True)
        for rel, n, rn in worst[:12]:
            print(f"   rel_l2={rel:.3e} |ref|={rn:.3e} {n}")
        print(f"   median rel {np.median([w[0] for w in worst]):.3e}", flush=True)


def timing():
    cfg = O.SwinIRConfig.classical_x4()
    sd = O.random_state_dict(cfg, 42, 1.5)
    m = build(cfg, sd, train=True)
    x = torch.rand(32, 3, 64, 64, device="cuda")
    t = torch.rand(32, 3, 256, 256, device="cuda")
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        out = m(x)
        torch.cuda.synchronize()
        t1 = time.time()
        loss = (out - t).abs().mean()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.time()
        for p in m.parameters():
            p.grad = None
        print(f"cfg3 bs32 it{it}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  loss {float(loss):.5f}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("tiny", "all"):
        tiny()
    if what in ("cfg2", "all"):
        big("cfg2")
    if what in ("cfg3", "all"):
        big("cfg3")
    if what in ("grads", "all"):
        grads()
    if what in ("timing", "all"):
        timing()


def infer_timing():
    for tag, cfg, bs, hw in (("cfg2 SwinIR-light x2", O.SwinIRConfig.light_x2(), 16, 48), ("cfg3 SwinIR x4 (fwd only)", O.SwinIRConfig.classical_x4(), 32, 64)):
        m = build(cfg, O.random_state_dict(cfg, 42, 1.0))
        x = torch.rand(bs, 3, hw, hw, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.time()
            n = 20
            for _ in range(n):
                y = m(x)
            torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        print(f"{tag}: bs{bs} {hw}x{hw} forward {1e3*dt:.2f} ms -> {bs * (hw * cfg.upscale) ** 2 / dt / 1e6:.1f} M HR px/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "infer":
    infer_timing()
