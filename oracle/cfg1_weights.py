"""Deterministic weights for MS_ResUNet fixtures (TEST INFRASTRUCTURE, used by oracle/make_golden.py and tests/ only).

The network has 24.9 M parameters: far too many to commit, so golden G11 stores only the seed.  Both sides -- the
reference model when the fixture is generated, the package's model when it is checked -- are filled from this one
generator, keyed by the state_dict's own names/shapes in order, and loaded with ``strict=True``.

He-style scaling keeps activations O(1) through the ~60 conv layers so that probes are meaningful: conv weights
N(0, 1/fan_in) (x0.25 on a residual unit's last conv), BatchNorm gamma U(0.6, 1.2), beta / running_mean N(0, 0.05),
running_var U(0.7, 1.3), conv biases N(0, 0.02).
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def fill_state_dict(template: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for k, v in template.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            t = torch.zeros(shape, dtype=v.dtype)
        elif k.endswith("running_var"):
            t = torch.rand(shape, generator=g) * 0.6 + 0.7
        elif k.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.05
        elif v.ndim == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            if "upCT" in k:                       # ConvTranspose2d: [in, out, kh, kw], stride 2 -> a quarter of the taps per output
                fan_in = shape[0] * shape[2] * shape[3] / 4
            gain = 0.25 if ("relu_varout" in k or "outvar_dimred" in k or k.endswith("conv3.weight")) else 1.0
            t = torch.randn(shape, generator=g) * (gain * math.sqrt(1.0 / fan_in))
        elif k.endswith(".weight"):               # BatchNorm gamma
            t = torch.rand(shape, generator=g) * 0.6 + 0.6
        else:                                     # biases (conv / BatchNorm beta)
            t = torch.randn(shape, generator=g) * (0.05 if ("bn" in k or "downsample" in k) else 0.02)
        out[k] = t.to(v.dtype)
    return out
