"""Test-set evaluation entry point with the reference's command line (modules/evaluate.py:54-234).

    python -m tpu_superresolution_amd.evaluate --scale X2 --data_root D --ckpt best_X2.pt [--save_dir preds ...]

Same ten flags, same pipeline and prints: eval transform (grayscale -> bicubic LR to HR size -> [0,1], :78) ->
``Shuffled2DPaired(split="test")`` -> min/max peek (:96-112) -> bicubic baseline PSNR/SSIM (:115-134) -> ``MS_ResUNet()``
+ checkpoint (``{"model": sd}`` or a raw state_dict, strict, :136-145) -> per-batch fp32 PSNR (:24-29) and SSIM,
non-finite guard (:172-178), optional bilinear resize to the HR size (:181-184), PNG dumps
``idx_%06d_{lr,hr,sr}.png`` under the policy --save_indices > --save_every/--save_start > first --save_n, always capped
by --save_n (:199-225) -> summary (:229-234).

BASELINE config 1 runs this on the CPU with stock torch operators (MS_ResUNet has no kernel in scope).
Additive: ``--arch swinir | hat | dat`` evaluates the MI355X SwinIR / HAT / DAT path (finetune_swinir.py model, RGB un-upscaled LR input, needs a
GPU + libsrk); ``main(argv)`` is callable from tests.  SSIM is ``metrics.ssim`` (restated, parity unpinned).
"""
from __future__ import annotations

import argparse
import re
import time
from pathlib import Path

import torch
from PIL import Image
from torch.utils.data import DataLoader

from .metrics import psnr, ssim
from .ms_resunet import MS_ResUNet
from .sr_datasets import Shuffled2DPaired
from .sr_transforms import build_pair_transform_eval


def save_tensor_as_png(x: torch.Tensor, path: Path, per_image_rescale: bool = False):
    """evaluate.py:31-51: [C,H,W] in [0,1] -> 8-bit PNG (clamp, or per-image min-max when asked; x255 then truncation, as
    torchvision's ToPILImage does for float tensors)."""
    x = x.detach().float().cpu()
    if per_image_rescale:
        lo, hi = float(x.min()), float(x.max())
        x = torch.zeros_like(x) if hi <= lo + 1e-8 else (x - lo) / (hi - lo)
    else:
        x = x.clamp(0.0, 1.0)
    a = x.mul(255).byte().numpy()
    img = Image.fromarray(a[0], mode="L") if a.shape[0] == 1 else Image.fromarray(a.transpose(1, 2, 0))
    img.save(str(path))


def _load_state(path: str):
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(ckpt, dict) and "model" in ckpt:
        return ckpt["model"], "[ckpt] loaded state_dict from 'model' key"
    if isinstance(ckpt, dict) and "params" in ckpt:
        return ckpt["params"], "[ckpt] loaded state_dict from 'params' key"
    return ckpt, "[ckpt] loaded raw state_dict"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=str, choices=["X2", "X4"], required=True, help="dataset configuration")
    ap.add_argument("--data_root", type=str, default="DeepRockSR-2D")
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--ckpt", type=str, required=True)
    ap.add_argument("--save_dir", type=str, default="preds")
    ap.add_argument("--save_n", type=int, default=16)
    ap.add_argument("--save_every", type=int, default=0, help="save every N-th sample by dataset index (0 = off)")
    ap.add_argument("--save_start", type=int, default=0, help="first index of the periodic saving (for save_every)")
    ap.add_argument("--save_indices", type=str, default="",
                    help="explicit comma-separated indices, e.g. '0,100,200'; takes priority over save_every")
    ap.add_argument("--arch", type=str, choices=["ms_resunet", "swinir", "hat", "dat"], default="ms_resunet")       # additive
    ap.add_argument("--device", type=str, default=None, help="additive: force 'cpu' / 'cuda' (default: cuda if available)")
    args = ap.parse_args(argv)

    device = torch.device(args.device) if args.device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    print("[device]", device, torch.cuda.get_device_name(0) if device.type == "cuda" else "-")
    swin = args.arch in ("swinir", "hat", "dat")
    if swin and device.type != "cuda":
        raise SystemExit(f"--arch {args.arch} runs on the MI355X HIP path only (no CPU fallback)")
    scale_int = 2 if args.scale.upper() == "X2" else 4

    if swin:
        from .sr_datasets import PairTransformValid
        tf_test = PairTransformValid(scale_int)
    else:
        tf_test = build_pair_transform_eval()
    test_ds = Shuffled2DPaired(args.data_root, split="test", scale=args.scale, transform_pair=tf_test)
    test_loader = DataLoader(test_ds, batch_size=args.batch_size, shuffle=False, num_workers=args.workers,
                             pin_memory=(device.type == "cuda"), persistent_workers=False)
    print(f"[data] test samples: {len(test_ds)} | steps: {len(test_loader)}")

    def upscaled(lr, hr):
        """What is compared with HR as the 'bicubic' prediction: the eval transform already upscaled LR for MS_ResUNet; for
        SwinIR (raw LR input) it is done here."""
        if lr.shape[-2:] == hr.shape[-2:]:
            return lr
        return torch.nn.functional.interpolate(lr, size=hr.shape[-2:], mode="bicubic", align_corners=False).clamp(0, 1)

    for lr, hr in test_loader:                          # _peek_batch, :96-112
        lf, hf = torch.isfinite(lr), torch.isfinite(hr)
        print("[peek] lr min/max:", float(lr[lf].min()) if lf.any() else float("nan"), float(lr[lf].max()) if lf.any() else float("nan"),
              "| hr min/max:", float(hr[hf].min()) if hf.any() else float("nan"), float(hr[hf].max()) if hf.any() else float("nan"),
              "| shapes:", tuple(lr.shape), tuple(hr.shape))
        break

    with torch.no_grad():                               # bicubic baseline, :115-134
        ps, ss = [], []
        for lr, hr in test_loader:
            lr, hr = lr.to(device, dtype=torch.float32), hr.to(device, dtype=torch.float32)
            up = upscaled(lr, hr)
            ps.append(psnr(up, hr, max_val=1.0))
            ss.append(float(ssim(up, hr, data_range=1.0, size_average=True)))
    print(f"[baseline] Bicubic PSNR: {sum(ps) / len(ps):.2f} dB | SSIM: {sum(ss) / len(ss):.4f}")

    if swin:
        from .finetune_swinir import build_sr_model
        model = build_sr_model(args.arch, scale_int, drop_path_rate=0.0)
    else:
        model = MS_ResUNet()
    state, msg = _load_state(args.ckpt)
    model.load_state_dict(state, strict=True)
    print(msg)
    model = model.to(device).eval()

    t0 = time.time()
    psnr_vals, ssim_vals = [], []
    out_dir = Path(args.save_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    saved = 0
    save_set = None
    if args.save_indices.strip():
        save_set = {int(x) for x in re.split(r"[,\s]+", args.save_indices.strip()) if x != ""}
        print(f"[save] explicit indices: {sorted(save_set)[:20]}{'...' if len(save_set) > 20 else ''}")
    elif args.save_every and args.save_every > 0:
        print(f"[save] every {args.save_every} samples starting at {args.save_start}")
    else:
        print(f"[save] first {args.save_n} samples (default mode)")

    global_idx = 0
    with torch.no_grad():
        for lr, hr in test_loader:
            lr, hr = lr.to(device, non_blocking=True), hr.to(device, non_blocking=True)
            with torch.amp.autocast("cuda", enabled=(device.type == "cuda" and not swin)):
                pred = model(lr)
                if not torch.isfinite(pred).all():
                    bad = (~torch.isfinite(pred)).float().mean().item()
                    raise RuntimeError(f"Pred has non-finite values: share={bad:.6f}, min={torch.nanmin(pred).item():.4g}, "
                                       f"max={torch.nanmax(pred).item():.4g}")
            if pred.shape[-2:] != hr.shape[-2:]:
                pred = torch.nn.functional.interpolate(pred, size=hr.shape[-2:], mode="bilinear", align_corners=False)
            pred_f, hr_f = pred.to(torch.float32), hr.to(torch.float32)
            psnr_vals.append(psnr(pred_f, hr_f, max_val=1.0))
            ssim_vals.append(float(ssim(pred_f, hr_f, data_range=1.0, size_average=True)))
            for b in range(pred.size(0)):
                idx = global_idx + b
                if save_set is not None:
                    want = idx in save_set
                elif args.save_every and args.save_every > 0:
                    want = idx >= args.save_start and (idx - args.save_start) % args.save_every == 0
                else:
                    want = saved < args.save_n
                if not want or saved >= args.save_n:       # --save_n caps every mode (:213-215)
                    continue
                stem = f"idx_{idx:06d}"
                save_tensor_as_png(lr[b], out_dir / f"{stem}_lr.png")
                save_tensor_as_png(hr[b], out_dir / f"{stem}_hr.png")
                save_tensor_as_png(pred[b], out_dir / f"{stem}_sr.png")
                saved += 1
            global_idx += pred.size(0)

    dt = time.time() - t0
    mean_psnr = sum(psnr_vals) / max(1, len(psnr_vals))
    mean_ssim = sum(ssim_vals) / max(1, len(ssim_vals))
    print(f"[done] test PSNR: {mean_psnr:.2f} dB | SSIM: {mean_ssim:.4f} | time: {dt:.1f}s for {len(test_ds)} samples")
    print(f"[saved] examples in: {out_dir.resolve()}")
    return {"psnr": mean_psnr, "ssim": mean_ssim, "bicubic_psnr": sum(ps) / len(ps), "bicubic_ssim": sum(ss) / len(ss),
            "saved": saved, "n": len(test_ds)}


if __name__ == "__main__":
    main()
