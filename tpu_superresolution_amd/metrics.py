"""Image-quality metrics used by the reference's scripts.

``batch_psnr``   train.py:46-56 / finetune_swinir.py:69-74   clamp to [0,1], 20 log10(max / sqrt(mse + 1e-8)), per image
``psnr``         evaluate.py:24-29                            no clamp, mse floored at 1e-10, mean over the batch (float)
``ssim``         ``pytorch_msssim.ssim`` (pin pytorch-msssim==1.0.0, sr_environment.yml:165; call sites train.py:169,
                 evaluate.py:127,195).  The package is not part of the reference tree and not installed here, so this is
                 a restatement of its published algorithm -- Wang et al. 2004 with an 11-tap Gaussian (sigma 1.5)
                 applied separably as a VALID (unpadded) depth-wise filter, K = (0.01, 0.03), per-channel map mean,
                 then mean over channels (and over the batch when size_average) -- **parity unpinned**: no
                 reference-produced fixture exists; tests check it against the closed form on cases with a known answer
                 (identical images -> 1, constant shift, scipy's gaussian_filter1d windows).

CPU tensors use the torch-operator forms below (evaluate.py / train.py run BASELINE cfg1 on the CPU).  fp32 GPU tensors go to
the libsrk kernels: ``psnr`` -> ``srk_eval_psnr``, ``batch_psnr`` -> ``srk_batch_psnr``, ``ssim`` (4-D,
H, W >= 11, default window) -> ``srk_ssim`` (csrc/metrics.hip, csrc/misc.hip): one fused pass, fixed-order sums, no per-pixel
intermediates in HBM.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _on_device(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


def batch_psnr(pred: torch.Tensor, target: torch.Tensor, max_val: float = 1.0) -> torch.Tensor:
    if _on_device(pred, target) and pred.size(0) <= 1024:
        from . import ops
        return ops.batch_psnr(pred.detach(), target.detach(), max_val)
    pred, target = pred.clamp(0.0, 1.0), target.clamp(0.0, 1.0)
    mse = ((pred - target) ** 2).reshape(pred.size(0), -1).mean(dim=1)
    return 20.0 * torch.log10(max_val / torch.sqrt(mse + 1e-8))


def psnr(x: torch.Tensor, y: torch.Tensor, max_val: float = 1.0) -> float:
    if _on_device(x, y) and x.size(0) <= 1024:
        from . import ops
        return float(ops.eval_psnr(x.detach(), y.detach(), max_val)[1])
    mse = torch.mean((x - y) ** 2, dim=[1, 2, 3]).clamp(min=1e-10)
    return float((20.0 * torch.log10(max_val / torch.sqrt(mse))).mean())


def gaussian_window(size: int = 11, sigma: float = 1.5, dtype=torch.float32, device=None) -> torch.Tensor:
    coords = torch.arange(size, dtype=torch.float32, device=device) - size // 2
    g = torch.exp(-(coords ** 2) / (2.0 * sigma ** 2))
    return (g / g.sum()).to(dtype)


def _blur_valid(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """Separable depth-wise VALID filtering of [B, C, H, W]; an axis shorter than the window is left unfiltered (the
    published implementation warns and skips it)."""
    C = x.shape[1]
    k = win.numel()
    if x.shape[2] >= k:
        x = F.conv2d(x, win.view(1, 1, k, 1).expand(C, 1, k, 1), groups=C)
    if x.shape[3] >= k:
        x = F.conv2d(x, win.view(1, 1, 1, k).expand(C, 1, 1, k), groups=C)
    return x


def ssim_torch(X: torch.Tensor, Y: torch.Tensor, data_range: float = 255.0, size_average: bool = True, win_size: int = 11,
               win_sigma: float = 1.5, K=(0.01, 0.03), nonnegative_ssim: bool = False) -> torch.Tensor:
    if X.shape != Y.shape:
        raise ValueError(f"Input images should have the same dimensions, but got {X.shape} and {Y.shape}.")
    if X.ndim != 4:
        raise ValueError(f"Input images should be 4-d tensors [B, C, H, W], but got {X.shape}")
    if win_size % 2 != 1:
        raise ValueError("Window size should be odd.")
    win = gaussian_window(win_size, win_sigma, X.dtype, X.device)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _blur_valid(X, win), _blur_valid(Y, win)
    s11 = _blur_valid(X * X, win) - mu1 * mu1
    s22 = _blur_valid(Y * Y, win) - mu2 * mu2
    s12 = _blur_valid(X * Y, win) - mu1 * mu2
    cs = (2 * s12 + C2) / (s11 + s22 + C2)
    smap = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs
    per_channel = smap.flatten(2).mean(-1)           # [B, C]
    if nonnegative_ssim:
        per_channel = torch.relu(per_channel)
    return per_channel.mean() if size_average else per_channel.mean(1)


def ssim(X: torch.Tensor, Y: torch.Tensor, data_range: float = 255.0, size_average: bool = True, **kw) -> torch.Tensor:
    """``pytorch_msssim.ssim`` signature (data_range default 255, as published)."""
    if (_on_device(X, Y) and not kw and X.ndim == 4 and X.shape == Y.shape and X.shape[2] >= 11 and X.shape[3] >= 11 and X.size(0) <= 1024
            and X.size(0) * X.size(1) < 65536):
        from . import ops
        per, mean = ops.ssim(X.detach(), Y.detach(), data_range)
        return mean.reshape(()) if size_average else per
    return ssim_torch(X, Y, data_range=data_range, size_average=size_average, **kw)
