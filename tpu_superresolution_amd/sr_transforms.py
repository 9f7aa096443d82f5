"""Paired (LR, HR) transforms of the reference (modules/sr_transforms.py:18-162), PIL + torch only.

The reference builds these on torchvision v2, which is not part of this image; for the inputs the pipelines see
(PIL images straight from ``Shuffled2DPaired``) every torchvision call reduces to a PIL call:

    T.Grayscale(1)(pil)                                  == pil.convert("L")
    TF.resize(pil, [H, W], BICUBIC, antialias=True)      == pil.resize((W, H), Image.BICUBIC)
    TF.crop / hflip / vflip (pil)                        == pil.crop / transpose(FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM)
    ToImage() + ToDtype(float32, scale=True)             == integer pixels / max of their dtype, [C, H, W]

Random draws keep the reference's generator and call order (``torch.randint`` for the crop corner: top then left,
:96-97; ``torch.rand(())`` for the horizontal then the vertical flip, :116-119), so a seeded run cuts the same patches.
Tensor inputs are accepted where the reference accepts them.  Restated from the source text: torchvision is absent,
so no reference-produced fixture exists for these (parity unpinned beyond the PIL identities above;
tests/test_cfg1_plumbing.py checks the identities' observable properties).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image


class PairCompose:
    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = list(transforms)

    def __call__(self, lr, hr):
        for t in self.transforms:
            lr, hr = t(lr, hr)
        return lr, hr


def _gray_tensor(x: torch.Tensor, num_output_channels: int) -> torch.Tensor:
    """torchvision rgb_to_grayscale on a [..., 3, H, W] tensor: ITU-R 601-2 luma, result in the input dtype."""
    if x.shape[-3] != 3:
        raise TypeError(f"Input image tensor permitted channel values are 1 or 3, but found {x.shape[-3]}")
    r, g, b = x.unbind(dim=-3)
    l = (0.2989 * r + 0.587 * g + 0.114 * b).to(x.dtype).unsqueeze(-3)
    return l.expand(*x.shape[:-3], 3, *x.shape[-2:]).contiguous() if num_output_channels == 3 else l


class PairGrayscale:
    """sr_transforms.py:26-52."""

    def __init__(self, num_output_channels: int = 1):
        if num_output_channels not in (1, 3):
            raise ValueError("num_output_channels should be either 1 or 3")
        self.n = num_output_channels

    def gray(self, x):
        if isinstance(x, torch.Tensor):
            if x.ndim == 2:
                return x.unsqueeze(0)
            if x.ndim == 3 and x.shape[0] == 1:
                return x
            return _gray_tensor(x, self.n)
        if isinstance(x, Image.Image):
            if x.mode in ("L", "F", "I", "I;16"):
                return x
            l = x.convert("L")
            return Image.merge("RGB", (l, l, l)) if self.n == 3 else l
        arr = np.asarray(x)
        if arr.ndim == 2:
            return arr
        return self.gray(Image.fromarray(arr))

    def __call__(self, lr, hr):
        return self.gray(lr), self.gray(hr)


class PairUpscaleLRtoHR:
    """Bicubic LR -> exactly the HR size (sr_transforms.py:55-63)."""

    def __call__(self, lr, hr):
        if lr.size != hr.size:                       # PIL: (W, H)
            lr = lr.resize((hr.size[0], hr.size[1]), Image.BICUBIC)
        return lr, hr


class PairRandomCrop:
    """The same random rectangle out of both images, which must already have one size (sr_transforms.py:65-111).
    Patch == image: untouched; patch larger than the image: centre crop to the smaller size."""

    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, lr, hr):
        th, tw = self.size
        if isinstance(hr, Image.Image):
            w, h = hr.size
        else:
            h, w = hr.shape[-2], hr.shape[-1]
        if h == th and w == tw:
            return lr, hr
        if h < th or w < tw:
            th, tw = min(th, h), min(tw, w)
            top, left = max(0, (h - th) // 2), max(0, (w - tw) // 2)
        else:
            top = int(torch.randint(0, h - th + 1, (1,)).item())
            left = int(torch.randint(0, w - tw + 1, (1,)).item())
        if isinstance(hr, Image.Image):
            box = (left, top, left + tw, top + th)
            return lr.crop(box), hr.crop(box)
        return lr[..., top:top + th, left:left + tw], hr[..., top:top + th, left:left + tw]


def _flip(x, horizontal: bool):
    if isinstance(x, Image.Image):
        return x.transpose(Image.FLIP_LEFT_RIGHT if horizontal else Image.FLIP_TOP_BOTTOM)
    return x.flip(-1 if horizontal else -2)


class PairFlips:
    """sr_transforms.py:112-122."""

    def __init__(self, p_flip=0.5, p_vflip=0.5):
        self.pf, self.pv = float(p_flip), float(p_vflip)

    def __call__(self, lr, hr):
        if torch.rand(()) < self.pf:
            lr, hr = _flip(lr, True), _flip(hr, True)
        if torch.rand(()) < self.pv:
            lr, hr = _flip(lr, False), _flip(hr, False)
        return lr, hr


_INT_MAX = {np.dtype(np.uint8): 255.0, np.dtype(np.uint16): 65535.0, np.dtype(np.int16): 32767.0,
            np.dtype(np.int32): 2147483647.0}


def to_tensor01(x) -> torch.Tensor:
    """ToImage + ToDtype(float32, scale=True): [C, H, W] float32; integer pixels are divided by their dtype's maximum."""
    if isinstance(x, torch.Tensor):
        t = x if x.ndim >= 3 else x.unsqueeze(0)
        if t.dtype.is_floating_point:
            return t.to(torch.float32)
        return t.to(torch.float32) / float(torch.iinfo(t.dtype).max)
    a = np.asarray(x)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8) * 255
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("="))
    a = np.ascontiguousarray(a)
    if a.dtype in _INT_MAX:
        t = torch.from_numpy(a.astype(np.float32)) / _INT_MAX[a.dtype]
    else:
        t = torch.from_numpy(a.astype(np.float32))
    return t.unsqueeze(0) if t.ndim == 2 else t.permute(2, 0, 1).contiguous()


class PairToTensor01:
    """sr_transforms.py:125-133."""

    def __call__(self, lr, hr):
        return to_tensor01(lr), to_tensor01(hr)


def build_pair_transform(do_flips: bool = True, patch_size: Optional[int] = None) -> PairCompose:
    """Training / validation pipeline (sr_transforms.py:136-152)."""
    stages = [PairGrayscale(), PairUpscaleLRtoHR()]
    if patch_size is not None:
        stages.append(PairRandomCrop(patch_size))
    if do_flips:
        stages.append(PairFlips())
    stages.append(PairToTensor01())
    return PairCompose(stages)


def build_pair_transform_eval() -> PairCompose:
    """Test pipeline (sr_transforms.py:154-162): grayscale -> bicubic LR to HR size -> [0, 1] tensors."""
    return PairCompose([PairGrayscale(), PairUpscaleLRtoHR(), PairToTensor01()])
