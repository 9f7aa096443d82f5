"""Generic trainer with the reference's command line (modules/train.py:183-429).

    python -m tpu_superresolution_amd.train --data_root D --scale X2 [--epochs 5 --scheduler OneCycle --loss l1 ...]

The reference hard-wires ``MS_ResUNet`` (:22, :279); so does the default here.  Same 19 flags and the same run shape:
seed (:214) -> worker/pin policy print (:217-220) -> loss select (:226-231) -> pair transforms (train: patch + flips,
valid: full image, :235-242) -> ``Shuffled2DPaired`` train/valid + loaders (valid batch = batch_size // 2, :245-273) ->
loader warm-up timing (:80-86, :276) -> model, ``--resume`` (``ckpt.get("model", ckpt)``, strict) / ``--finetune`` /
``--freeze_regex`` / ``--ft_lr`` (:282-300) -> AdamW -> OneCycle (per batch: pct_start 0.1, cos, div 10, final div 100) |
Exponential (gamma = 0.5^(1/50), per epoch) | None (:308-333) -> optimizer / scheduler restore unless fine-tuning
(:335-342) -> fp16 autocast + GradScaler on the GPU (:345) -> per epoch: train (NaN guard :110-113, clip 1.0 :119),
validate (nan_to_num, loss, PSNR :46-56, SSIM), epoch print (:375-381), ETA print (:387-391), ``best_{scale}.pt`` =
``{"model","opt","sched","epoch","args"}`` on the best validation loss (:393-401) -> ``loss_curve_{scale}.png``
(:403-419) -> total-time prints.

MS_ResUNet runs on stock torch operators (CPU, or the GPU through torch's ROCm operators): SURVEY 8 row a17 puts no
kernel of it in scope.  Additive: ``--arch swinir | hat | dat`` trains the MI355X SwinIR / HAT / DAT path with this loop (RGB LR patches of
``--patch_size``, no autocast -- bf16 MFMA is inside the kernels -- torch AdamW on the flat-buffer views; the tuned loop
for it is ``finetune_swinir.py``); ``main(argv)`` is callable from tests.  SSIM is ``metrics.ssim`` (restated from the
published pytorch-msssim algorithm; parity unpinned).
"""
from __future__ import annotations

import argparse
import math
import os
import random
import re
import time
from datetime import timedelta

import torch
import torch.nn.functional as F
import torch.optim as optim
from torch.optim.lr_scheduler import OneCycleLR
from torch.utils.data import DataLoader, IterableDataset

from .metrics import batch_psnr, ssim as ssim_ms
from .ms_resunet import MS_ResUNet
from .sr_datasets import Shuffled2DPaired
from .sr_transforms import build_pair_transform


def fmt(seconds: float) -> str:
    return str(timedelta(seconds=int(seconds)))


def seed_everything(seed: int = 42):
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = False
    torch.backends.cudnn.benchmark = True          # MIOpen find-mode on ROCm: faster for fixed sizes


def L1_loss(pred, target):
    return F.l1_loss(pred, target)


def mse_loss(pred, target):
    return F.mse_loss(pred, target)


def make_loader(ds, batch_size, workers, pin=True, shuffle=False, drop_last=False, persistent=False):
    iterable = isinstance(ds, IterableDataset)
    kw = dict(dataset=ds, batch_size=batch_size, shuffle=(False if iterable else shuffle), drop_last=drop_last,
              num_workers=workers, pin_memory=pin)
    if workers and workers > 0 and not iterable:
        kw["persistent_workers"] = persistent
        kw["prefetch_factor"] = 2
    return DataLoader(**kw)


@torch.no_grad()
def warmup_profile(dl, n_batches=3):
    t0 = time.time()
    for i, _ in enumerate(dl):
        if i == 0:
            print(f"[profile] first batch load: {time.time() - t0:.2f}s")
        if i + 1 >= n_batches:
            break
    print(f"[profile] {n_batches} batches load: {time.time() - t0:.2f}s")


def train_one_epoch(model, loader, optimizer, scaler, device, epoch, loss_fn, sched, is_batch_sched, autocast=True):
    model.train()
    data_t = step_t = total_loss = 0.0
    n_steps, end = 0, time.time()
    for lr, hr in loader:
        data_time = time.time() - end
        lr, hr = lr.to(device, non_blocking=True), hr.to(device, non_blocking=True)
        t0 = time.time()
        optimizer.zero_grad(set_to_none=True)
        with torch.amp.autocast("cuda", enabled=(autocast and device.type == "cuda")):
            out = model(lr)
            if not torch.isfinite(out).all():
                raise RuntimeError("Model produced NaN/Inf: lower max_lr, check residual_scale/init")
            loss = loss_fn(out, hr)
        scaler.scale(loss).backward()
        scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        scaler.step(optimizer)
        if is_batch_sched and sched is not None:
            sched.step()
        scaler.update()
        total_loss += loss.item()
        data_t += data_time
        step_t += time.time() - t0
        n_steps += 1
        end = time.time()
    n = max(1, n_steps)
    return total_loss / n, data_t / n, step_t / n


@torch.no_grad()
def validate(model, loader, device, loss_fn, autocast=True):
    model.eval()
    tot = sum_psnr = sum_ssim = 0.0
    n = n_imgs = 0
    for it, (lr, hr) in enumerate(loader):
        if it == 0:
            print(f"[val] batch0 shapes: lr={tuple(lr.shape)}, hr={tuple(hr.shape)}, numel(lr)={lr.numel()}, numel(hr)={hr.numel()}")
        lr, hr = lr.to(device, non_blocking=True), hr.to(device, non_blocking=True)
        with torch.amp.autocast("cuda", enabled=(autocast and device.type == "cuda")):
            lr = torch.nan_to_num(lr, nan=0.0, posinf=0.0, neginf=0.0)
            hr = torch.nan_to_num(hr, nan=0.0, posinf=0.0, neginf=0.0)
            out = model(lr)
            loss = loss_fn(out, hr)
        sum_psnr += batch_psnr(out, hr).sum().item()
        sum_ssim += ssim_ms(out.clamp(0, 1).float(), hr.clamp(0, 1).float(), data_range=1.0).item() * lr.size(0)
        n_imgs += lr.size(0)
        tot += loss.item()
        n += 1
    return tot / max(1, n), sum_psnr / max(1, n_imgs), sum_ssim / max(1, n_imgs)


def _save_loss_curve(train_hist, val_hist, scale: str) -> str:
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    epochs = list(range(1, len(train_hist) + 1))
    path = f"loss_curve_{scale}.png"
    plt.figure(figsize=(8, 5))
    plt.plot(epochs, train_hist, label="train loss")
    plt.plot(epochs, val_hist, label="val loss")
    plt.xlabel("Epoch")
    plt.ylabel("Loss")
    plt.title(f"Train vs Val loss ({scale})")
    plt.grid(True)
    plt.legend()
    plt.tight_layout()
    plt.savefig(path, dpi=150)
    plt.close()
    return path


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_root", type=str, required=True)
    ap.add_argument("--scale", type=str, choices=["X2", "X4"], required=True, help="training scale")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--scheduler", type=str, choices=["OneCycle", "Exponential", "None"], default="None")
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--loss", type=str, choices=["mse", "l1"], default="mse")
    ap.add_argument("--patch_size", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--weight_decay", type=float, default=0)
    ap.add_argument("--no_flips", action="store_true")
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no_pin", action="store_true")
    ap.add_argument("--no_persistent", action="store_true")
    ap.add_argument("--time_log_every", type=int, default=10)
    ap.add_argument("--resume", type=str, default=None, help="path of a .pt with a 'model' key to continue from")
    ap.add_argument("--finetune", action="store_true",
                    help="use the checkpoint as initialisation only (optimizer / scheduler are not restored)")
    ap.add_argument("--freeze_regex", type=str, default=None, help="regex over parameter names to freeze, e.g. 'inc|down'")
    ap.add_argument("--ft_lr", type=float, default=None, help="separate learning rate for fine-tuning (default: --lr)")
    ap.add_argument("--arch", type=str, choices=["ms_resunet", "swinir", "hat", "dat"], default="ms_resunet")       # additive
    ap.add_argument("--device", type=str, default=None, help="additive: force 'cpu' / 'cuda' (default: cuda if available)")
    args = ap.parse_args(argv)

    seed_everything(args.seed)
    t_all_start = time.time()
    if args.workers is None:
        cpu = os.cpu_count() or 4
        args.workers = min(2, cpu) if os.name == "nt" else min(8, max(2, cpu // 2))
    print(f"[cfg] workers={args.workers}, pin={not args.no_pin}, persistent={not args.no_persistent}")
    device = torch.device(args.device) if args.device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    print("[device]", device, torch.cuda.get_device_name(0) if device.type == "cuda" else "-")
    swin = args.arch in ("swinir", "hat", "dat")          # the transformer models of modules/ on the HIP path: RGB LR patches in, x scale out
    if swin and device.type != "cuda":
        raise SystemExit(f"--arch {args.arch} runs on the MI355X HIP path only (no CPU fallback)")
    scale_int = 2 if args.scale.upper() == "X2" else 4
    pin = (not args.no_pin) and device.type == "cuda"

    loss_fn = {"mse": mse_loss, "l1": L1_loss}[args.loss]
    if swin:
        from .sr_datasets import PairTransformTrain, PairTransformValid
        tf_train, tf_valid = PairTransformTrain(args.patch_size, scale_int), PairTransformValid(scale_int)
    else:
        tf_train = build_pair_transform(patch_size=args.patch_size, do_flips=not args.no_flips)
        tf_valid = build_pair_transform(do_flips=False)            # validation on full images
    train_ds = Shuffled2DPaired(args.data_root, split="train", scale=args.scale, transform_pair=tf_train)
    valid_ds = Shuffled2DPaired(args.data_root, split="valid", scale=args.scale, transform_pair=tf_valid)
    train_loader = make_loader(train_ds, args.batch_size, args.workers, pin=pin, shuffle=True, drop_last=False,
                               persistent=not args.no_persistent)
    valid_loader = make_loader(valid_ds, max(1, args.batch_size // 2), args.workers, pin=pin, shuffle=False, drop_last=False,
                               persistent=not args.no_persistent)
    print(f"\n[profile {args.scale} loader]")
    warmup_profile(train_loader, n_batches=3)

    if swin:
        from .finetune_swinir import build_sr_model
        model = build_sr_model(args.arch, scale_int).to(device)
    else:
        model = MS_ResUNet().to(device)
    ckpt = None
    if args.resume is not None:
        ckpt = torch.load(args.resume, map_location=device, weights_only=True)
        model.load_state_dict(ckpt.get("model", ckpt), strict=True)
        print(f"[ckpt] loaded model weights from {args.resume}")
    if args.finetune and args.freeze_regex is not None:
        pattern = re.compile(args.freeze_regex)
        for name, p in model.named_parameters():
            if pattern.search(name):
                p.requires_grad = False
        print(f"[finetune] froze params matching regex: {args.freeze_regex}")
    trainable = [p for p in model.parameters() if p.requires_grad]
    lr = args.ft_lr if (args.finetune and args.ft_lr is not None) else args.lr
    opt = optim.AdamW(trainable, lr=lr, weight_decay=args.weight_decay)

    sched, is_batch_sched = None, False
    if args.scheduler == "OneCycle":
        sched = OneCycleLR(optimizer=opt, max_lr=lr, steps_per_epoch=len(train_loader), epochs=args.epochs, pct_start=0.1,
                           anneal_strategy="cos", div_factor=10, final_div_factor=100)
        is_batch_sched = True
    elif args.scheduler == "Exponential":
        gamma = 0.5 ** (1.0 / 50)                                   # halve every 50 epochs
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma)
        print(f"[sched] ExponentialLR: gamma={gamma:.6f}")
    if ckpt is not None and not args.finetune:
        if ckpt.get("opt") is not None:
            opt.load_state_dict(ckpt["opt"])
            print("[ckpt] restored optimizer state")
        if sched is not None and ckpt.get("sched") is not None:
            sched.load_state_dict(ckpt["sched"])
            print("[ckpt] restored scheduler state")
    use_amp = device.type == "cuda" and not swin
    scaler = torch.amp.GradScaler("cuda", enabled=use_amp)

    best, t_start = math.inf, time.time()
    hist_train, hist_val = [], []
    for epoch in range(1, args.epochs + 1):
        t_ep = time.time()
        tr_loss, d_t, b_t = train_one_epoch(model, train_loader, opt, scaler, device, epoch, loss_fn, sched, is_batch_sched,
                                            autocast=use_amp)
        t_tr = time.time() - t_ep
        t_v = time.time()
        val_loss, val_psnr, val_ssim = validate(model, valid_loader, device, loss_fn, autocast=use_amp)
        t_val = time.time() - t_v
        hist_train.append(tr_loss)
        hist_val.append(val_loss)
        print(f"[{args.scale}] epoch {epoch}: train_loss {tr_loss:.7f}, val_loss {val_loss:.7f} | "
              f"val_PSNR {val_psnr:.2f} dB, val_SSIM {val_ssim:.4f} | (data {d_t:.3f}/batch {b_t:.3f}) | "
              f"time: train {t_tr:.1f}s, val {t_val:.1f}s, total {time.time() - t_ep:.1f}s")
        if not is_batch_sched and sched is not None:
            sched.step()
        if args.time_log_every and (epoch % args.time_log_every == 0 or epoch == 1):
            elapsed = time.time() - t_start
            print(f"[{args.scale}][time] elapsed={fmt(elapsed)} | avg/epoch={fmt(elapsed / epoch)} | "
                  f"ETA≈{fmt(elapsed / epoch * (args.epochs - epoch))}")
        if val_loss < best:
            best = val_loss
            torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "opt": opt.state_dict(),
                        "sched": sched.state_dict() if sched is not None else None, "epoch": epoch, "args": vars(args)},
                       f"best_{args.scale}.pt")
    fig_path = _save_loss_curve(hist_train, hist_val, args.scale)
    print(f"[plot] saved loss curves to {fig_path}")
    print(f"[{args.scale}][time] total={fmt(time.time() - t_start)}")
    print(f"[ALL][time] total train time={fmt(time.time() - t_all_start)}")
    return {"train_loss": hist_train, "val_loss": hist_val, "best": best}


if __name__ == "__main__":
    main()
