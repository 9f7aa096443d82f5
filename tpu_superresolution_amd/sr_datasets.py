"""Paired LR/HR PNG dataset of the reference (modules/sr_datasets.py:14-73), host side, PIL only.

Directory contract (DeepRockSR-2D "shuffled2D"):
    <root>/shuffled2D/shuffled2D_<split>_HR/*.png
    <root>/shuffled2D/shuffled2D_<split>_LR_default_<X2|X4>/*x2.png  (stem = HR stem + optional [_-]x<k>)
Pairs are matched by stem; ``transform_pair(lr_pil, hr_pil) -> (lr_tensor, hr_tensor)``.
"""
from __future__ import annotations

import random
import re
from pathlib import Path
from typing import Callable, Optional, Tuple

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def _dirs(root: str, split: str, scale: str) -> Tuple[Path, Path]:
    base = Path(root) / "shuffled2D"
    hr, lr = base / f"shuffled2D_{split}_HR", base / f"shuffled2D_{split}_LR_default_{scale}"
    if not (hr.exists() and lr.exists()):
        raise FileNotFoundError(f"HR/LR directories not found for split={split}, scale={scale} under {base}")
    return hr, lr


def _strip_lr_suffix(stem: str, scale: str) -> str:
    suf = scale.lower()
    if not suf.startswith("x"):
        suf = "x" + suf
    return re.sub(rf"([_-]?){re.escape(suf)}$", "", stem, flags=re.IGNORECASE)


class Shuffled2DPaired(Dataset):
    def __init__(self, root: str, split: str = "train", scale: str = "X2",
                 exts: Tuple[str, ...] = (".png", ".jpg", ".jpeg", ".tif", ".tiff"), transform_pair: Optional[Callable] = None):
        self.hr_dir, self.lr_dir = _dirs(root, split, scale)
        self.transform_pair = transform_pair
        hr_map = {p.stem: p for p in sorted(self.hr_dir.iterdir()) if p.suffix.lower() in exts}
        if not hr_map:
            raise RuntimeError(f"no HR files in {self.hr_dir}")
        self.pairs = []
        for p in sorted(self.lr_dir.iterdir()):
            if p.suffix.lower() in exts:
                hr = hr_map.get(_strip_lr_suffix(p.stem, scale))
                if hr is not None:
                    self.pairs.append((p, hr))
        if not self.pairs:
            raise RuntimeError("no LR/HR pairs with matching file stems")

    def __len__(self):
        return len(self.pairs)

    @staticmethod
    def _open(p: Path) -> Image.Image:
        with Image.open(p) as img:
            return img.copy()

    def __getitem__(self, idx: int):
        lr_path, hr_path = self.pairs[idx]
        lr, hr = self._open(lr_path), self._open(hr_path)
        if self.transform_pair is not None:
            lr, hr = self.transform_pair(lr, hr)
        return lr, hr


# ---- minimal paired transforms of finetune_swinir.py:80-131 (no augmentation) -----------------------
def pil_to_tensor01(img: Image.Image) -> torch.Tensor:
    """uint8 PIL -> float32 [C,H,W] in [0,1] (torchvision ToImage + ToDtype(scale=True) for 8-bit inputs)."""
    a = np.asarray(img)
    if a.dtype == np.uint16:
        t = torch.from_numpy(a.astype(np.float32) / 65535.0)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a).astype(np.float32) / 255.0)
    return t.unsqueeze(0) if t.ndim == 2 else t.permute(2, 0, 1).contiguous()


def ensure_3ch(t: torch.Tensor) -> torch.Tensor:
    if t.ndim != 3:
        raise ValueError(f"Expected [C,H,W], got {tuple(t.shape)}")
    if t.size(0) == 1:
        return t.repeat(3, 1, 1)
    if t.size(0) != 3:
        raise ValueError(f"Expected C=1 or C=3, got C={t.size(0)}")
    return t


def paired_random_crop(lr_t: torch.Tensor, hr_t: torch.Tensor, lr_patch: int, scale: int):
    """LR crop at (top, left), HR crop at (top*scale, left*scale)  (finetune_swinir.py:96-110)."""
    _, h, w = lr_t.shape
    if h < lr_patch or w < lr_patch:
        raise ValueError(f"LR image too small for patch {lr_patch}: lr_size=({h},{w})")
    top, left = random.randint(0, h - lr_patch), random.randint(0, w - lr_patch)
    hp = lr_patch * scale
    return (lr_t[:, top:top + lr_patch, left:left + lr_patch],
            hr_t[:, top * scale:top * scale + hp, left * scale:left * scale + hp])


class PairTransformTrain:
    def __init__(self, lr_patch: int, scale: int):
        self.lr_patch, self.scale = lr_patch, scale

    def __call__(self, lr_pil, hr_pil):
        lr, hr = ensure_3ch(pil_to_tensor01(lr_pil)), ensure_3ch(pil_to_tensor01(hr_pil))
        return paired_random_crop(lr, hr, self.lr_patch, self.scale)


class PairTransformValid:
    def __init__(self, scale: int):
        self.scale = scale

    def __call__(self, lr_pil, hr_pil):
        return ensure_3ch(pil_to_tensor01(lr_pil)), ensure_3ch(pil_to_tensor01(hr_pil))


# ---- device-resident training set (first slice) ---------------------------------------------------------
class DevicePairPool:
    """Pre-decoded 8-bit LR/HR pairs in GPU memory + the paired train transform as one kernel pair per batch
    (`srk_paired_crop_u8`).  `sample(indices)` draws the crop corners with the same two `random.randint` calls per sample, in
    the same order, as `paired_random_crop` (finetune_swinir.py:96-110), so a host pipeline and this pool produce identical
    batches from the same `random` state.  16-bit images are not supported here (use the host transform)."""

    def __init__(self, pairs, lr_patch: int, scale: int, device="cuda"):
        """pairs: iterable of (lr, hr) PIL images or uint8 arrays [H,W] / [H,W,1|3]."""
        self.lr_patch, self.scale, self.device = int(lr_patch), int(scale), torch.device(device)
        chunks, self.meta, off = [], [], 0
        for lr, hr in pairs:
            entry = []
            for img in (lr, hr):
                a = np.ascontiguousarray(np.asarray(img))
                if a.dtype != np.uint8:
                    raise ValueError(f"DevicePairPool holds 8-bit images only, got {a.dtype}")
                if a.ndim == 2:
                    a = a[:, :, None]
                if a.ndim != 3 or a.shape[2] not in (1, 3):
                    raise ValueError(f"Expected C=1 or C=3, got shape {a.shape}")
                entry.append((off, a.shape[0], a.shape[1], a.shape[2]))
                chunks.append(a.reshape(-1))
                off += a.size
            (_, lh, lw, _), (_, hh, hw, _) = entry
            if lh < self.lr_patch or lw < self.lr_patch:
                raise ValueError(f"LR image too small for patch {self.lr_patch}: lr_size=({lh},{lw})")
            if hh < lh * self.scale or hw < lw * self.scale:
                raise ValueError(f"HR image ({hh},{hw}) smaller than scale x LR ({lh},{lw})")
            self.meta.append(tuple(entry))
        if not self.meta:
            raise ValueError("DevicePairPool: no images")
        self.pool = torch.from_numpy(np.concatenate(chunks)).to(self.device)

    def __len__(self):
        return len(self.meta)

    def sample(self, indices):
        """-> (lr [B,3,P,P], hr [B,3,P*s,P*s]) fp32 on the device; advances the global `random` state like the host transform."""
        from ._lib import check, lib
        P, s = self.lr_patch, self.scale
        ld, hd = [], []
        for i in indices:
            (lo, lh, lw, lc), (ho, hh, hw, hc) = self.meta[int(i)]
            top, left = random.randint(0, lh - P), random.randint(0, lw - P)
            ld.append((lo, lh, lw, lc, top, left))
            hd.append((ho, hh, hw, hc, top * s, left * s))
        B = len(ld)
        desc = torch.tensor(ld + hd, dtype=torch.int64).to(self.device)
        lr = torch.empty(B, 3, P, P, dtype=torch.float32, device=self.device)
        hr = torch.empty(B, 3, P * s, P * s, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(lib().srk_paired_crop_u8(self.pool.data_ptr(), desc[:B].data_ptr(), desc[B:].data_ptr(), lr.data_ptr(), hr.data_ptr(),
                                       B, P, s, st))
        return lr, hr
