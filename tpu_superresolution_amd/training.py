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


class GraphedTrainStep:
    """The whole train step of a host-orchestrated model (HAT / DAT: hundreds of C-ABI launches plus the small torch ops between
    them, launch-bound on the host) captured ONCE into a hipGraph and replayed: forward (activations kept, DropPath factors drawn by
    the graph-safe generator, BatchNorm running statistics updated in place), L1, the hand-written backward, gradient clipping and
    the optimizer step.

        step = GraphedTrainStep(model, torch.optim.AdamW(model.parameters(), lr=2e-5, capturable=True), max_grad_norm=1.0)
        loss, bad = step(lr_batch, hr_batch)        # copies the batch into the graph's static inputs and replays

    Requirements: static batch shape, an optimizer constructed with ``capturable=True``, no gradient all-reduce hook on the model
    (world size 1: collectives stay outside graphs here), nothing in the step that reads a device value on the host.  The first call
    runs ``warmup`` eager steps (kernel attributes, caches and workspaces get set up outside the capture), then captures.  DropPath
    factors are drawn OUTSIDE the graph, before every replay, into a static buffer the captured forward reads (``model.draw_drop_path``
    / ``model._drop_override``): a fresh draw per step does not depend on how the graph-captured generator advances."""

    def __init__(self, model, optimizer, max_grad_norm: float = 1.0, warmup: int = 2):
        if getattr(model, "grad_sync", None) is not None:
            raise ValueError("GraphedTrainStep: detach the gradient synchronizer (graph capture is for single-process steps)")
        self.model, self.opt, self.max_grad_norm, self.warmup = model, optimizer, float(max_grad_norm), int(warmup)
        self.graph = None
        self.x = self.t = self.loss = self.bad = self.drop = None

    def _eager(self, x, t):
        self.opt.zero_grad(set_to_none=True)
        loss, bad = l1_loss_checked(self.model(x), t)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.opt.step()
        return loss.detach(), bad

    def __call__(self, lr_img: torch.Tensor, hr_img: torch.Tensor):
        draw = getattr(self.model, "draw_drop_path", None)
        if self.graph is None:
            self.x, self.t = lr_img.clone(), hr_img.clone()
            if draw is not None and self.model.training:
                d = draw(lr_img.shape[0], lr_img.device)
                if d is not None:
                    self.drop = d.clone()
                    self.model._drop_override = self.drop        # the captured forward reads this buffer
            side = torch.cuda.Stream(device=lr_img.device)
            side.wait_stream(torch.cuda.current_stream(lr_img.device))
            with torch.cuda.stream(side):              # warm-up on a side stream, as torch.cuda.graphs asks
                for _ in range(max(self.warmup, 1)):
                    self._eager(self.x, self.t)
            torch.cuda.current_stream(lr_img.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss, self.bad = self._eager(self.x, self.t)
        elif lr_img.shape != self.x.shape or hr_img.shape != self.t.shape:
            raise ValueError(f"GraphedTrainStep was captured for {tuple(self.x.shape)} -> {tuple(self.t.shape)}")
        self.x.copy_(lr_img)
        self.t.copy_(hr_img)
        if self.drop is not None:
            self.drop.copy_(draw(lr_img.shape[0], lr_img.device))
        self.graph.replay()
        return self.loss, self.bad

    def close(self) -> None:
        """Detach from the model (its forward draws its own DropPath factors again)."""
        if getattr(self.model, "_drop_override", None) is self.drop:
            self.model._drop_override = None
