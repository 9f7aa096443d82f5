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
    batches from the same `random` state.  8-bit and 16-bit (uint16 -> value / 65535, as pil_to_tensor01) images, mixed freely.

    ``shard_bytes``: when the decoded set is larger than this, it stays in PINNED host memory as shards and only two shards
    live on the device: ``prefetch(s)`` starts the asynchronous copy of shard s on a side stream, ``sample`` of an index in a
    shard that is not resident switches to it (joining its copy) and prefetches the next one -- the host->device transfer of
    shard s+1 overlaps the training steps on shard s.  ``shard_of(i)`` tells a sampler which shard an image
    lives in, so that epochs can be ordered shard by shard."""

    def __init__(self, pairs, lr_patch: int, scale: int, device="cuda", shard_bytes: Optional[int] = None):
        """pairs: iterable of (lr, hr) PIL images or uint8 / uint16 arrays [H,W] / [H,W,1|3]."""
        self.lr_patch, self.scale, self.device = int(lr_patch), int(scale), torch.device(device)
        shards, chunks, self.meta, off = [], [], [], 0
        for lr, hr in pairs:
            entry, pieces, size = [], [], 0
            for img in (lr, hr):
                a = np.asarray(img)
                if a.dtype.byteorder == ">":
                    a = a.astype(a.dtype.newbyteorder("="))
                a = np.ascontiguousarray(a)
                if a.dtype not in (np.uint8, np.uint16):
                    raise ValueError(f"DevicePairPool holds 8-bit and 16-bit unsigned images, got {a.dtype}")
                if a.ndim == 2:
                    a = a[:, :, None]
                if a.ndim != 3 or a.shape[2] not in (1, 3):
                    raise ValueError(f"Expected C=1 or C=3, got shape {a.shape}")
                wide = int(a.dtype == np.uint16)
                size += size & 1 if wide else 0                   # uint16 samples start at an even byte
                entry.append([size, a.shape[0], a.shape[1], a.shape[2] | (wide << 8)])
                pieces.append((size, a.view(np.uint8).reshape(-1)))
                size += a.nbytes
            (_, lh, lw, _), (_, hh, hw, _) = entry
            if lh < self.lr_patch or lw < self.lr_patch:
                raise ValueError(f"LR image too small for patch {self.lr_patch}: lr_size=({lh},{lw})")
            if hh < lh * self.scale or hw < lw * self.scale:
                raise ValueError(f"HR image ({hh},{hw}) smaller than scale x LR ({lh},{lw})")
            off += off & 1
            if shard_bytes and chunks and off + size > shard_bytes:
                shards.append((chunks, off))
                chunks, off = [], 0
            for e, (rel, flat) in zip(entry, pieces):
                e[0] = off + rel
            chunks.append((off, pieces))
            self.meta.append((len(shards),) + tuple(tuple(e) for e in entry))
            off += size
        if not self.meta:
            raise ValueError("DevicePairPool: no images")
        shards.append((chunks, off))
        self._host = []
        for sh_chunks, total in shards:
            buf = np.zeros(total, dtype=np.uint8)
            for base, pieces in sh_chunks:
                for rel, flat in pieces:
                    buf[base + rel:base + rel + flat.size] = flat
            t = torch.from_numpy(buf)
            self._host.append(t.pin_memory() if (self.device.type == "cuda" and len(shards) > 1) else t)
        self.num_shards = len(self._host)
        self._resident = {}                   # shard index -> (device tensor, ready event or None)
        self._side = torch.cuda.Stream(device=self.device) if (self.device.type == "cuda" and self.num_shards > 1) else None
        self.pool = self._host[0].to(self.device)          # single-shard pools: the whole set on the device (as before)
        self._resident[0] = (self.pool, None)
        self._current = 0

    def __len__(self):
        return len(self.meta)

    def shard_of(self, index: int) -> int:
        return self.meta[int(index)][0]

    def prefetch(self, shard: int) -> None:
        """Start the asynchronous host->device copy of a shard on the side stream (pinned source: a true async DMA)."""
        shard = int(shard) % self.num_shards
        if shard in self._resident or self._side is None:
            return
        for old in [k for k in self._resident if k != self._current]:      # keep at most two shards on the device
            del self._resident[old]
        with torch.cuda.stream(self._side):
            dev = self._host[shard].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._resident[shard] = (dev, ev)

    def _use(self, shard: int) -> torch.Tensor:
        if shard not in self._resident:
            self.prefetch(shard)
        dev, ev = self._resident[shard]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._resident[shard] = (dev, None)
        if shard != self._current:
            self._current = shard
            self.prefetch(shard + 1)
        self.pool = dev
        return dev

    def sample(self, indices):
        """-> (lr [B,3,P,P], hr [B,3,P*s,P*s]) fp32 on the device; advances the global `random` state like the host transform."""
        from ._lib import check, lib
        P, s = self.lr_patch, self.scale
        ld, hd = [], []
        shard_ids = {self.meta[int(i)][0] for i in indices}
        if len(shard_ids) != 1:
            raise ValueError("a batch must come from one shard (order the epoch with shard_of())")
        pool = self._use(shard_ids.pop())
        for i in indices:
            _, (lo, lh, lw, lc), (ho, hh, hw, hc) = self.meta[int(i)]
            top, left = random.randint(0, lh - P), random.randint(0, lw - P)
            ld.append((lo, lh, lw, lc, top, left))
            hd.append((ho, hh, hw, hc, top * s, left * s))
        B = len(ld)
        desc = torch.tensor(ld + hd, dtype=torch.int64).to(self.device)
        lr = torch.empty(B, 3, P, P, dtype=torch.float32, device=self.device)
        hr = torch.empty(B, 3, P * s, P * s, dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(lib().srk_paired_crop_u8(pool.data_ptr(), desc[:B].data_ptr(), desc[B:].data_ptr(), lr.data_ptr(), hr.data_ptr(),
                                       B, P, s, st))
        return lr, hr
