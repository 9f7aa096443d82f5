"""Training-step pieces of finetune_swinir.py:148-179 on the HIP path: fused L1 loss (+ finite check),
one-call train step."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops


class _L1Loss(torch.autograd.Function):
    """F.l1_loss(pred, target) (finetune_swinir.py:66-67): forward and d(pred) come out of one kernel pass,
    which also counts non-finite predictions (assert_finite, :133-143) without a host sync."""

    @staticmethod
    def forward(ctx, pred, target):
        loss, d_pred, bad = ops.l1_loss_fwd_bwd(pred.contiguous(), target.contiguous().float(), want_grad=True)
        ctx.save_for_backward(d_pred)
        ctx.mark_non_differentiable(bad)
        return loss.reshape(()), bad

    @staticmethod
    def backward(ctx, g_loss, _g_bad):
        (d_pred,) = ctx.saved_tensors
        return d_pred * g_loss, None


def l1_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _L1Loss.apply(pred, target)[0]


def l1_loss_checked(pred: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (loss, nonfinite_count) both device tensors."""
    return _L1Loss.apply(pred, target)


def train_step(model, optimizer, lr_img: torch.Tensor, hr_img: torch.Tensor, sync=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One step of the reference loop (finetune_swinir.py:154-176): zero_grad, forward, L1, backward,
    [gradient all-reduce], clip + AdamW.  Returns (loss, nonfinite_count) as device tensors (no host sync)."""
    optimizer.zero_grad(set_to_none=True)
    out = model(lr_img)
    loss, bad = l1_loss_checked(out, hr_img)
    loss.backward()
    if sync is not None:
        sync.finish()
    if hasattr(optimizer, "max_grad_norm"):      # FusedAdamW: the device-side counter gates the update (no host sync)
        optimizer.step(nonfinite=bad)
    else:
        optimizer.step()
    return loss.detach(), bad


def assert_finite_step(loss: torch.Tensor, bad: torch.Tensor) -> None:
    """Host-side check with the reference's error behaviour (RuntimeError on non-finite output / loss)."""
    nb = int(bad)
    if nb:
        raise RuntimeError(f"out has non-finite values: count={nb}")
    if not bool(torch.isfinite(loss)):
        raise RuntimeError("loss has non-finite values")
