"""Fused global-norm clip + AdamW over the engine's flat fp32 buffers (libsrk kernels).

Semantics = ``torch.nn.utils.clip_grad_norm_(params, max_norm)`` followed by ``torch.optim.AdamW.step()``
as used by the reference training loop (finetune_swinir.py:168-171, :303), without the host sync the
reference's clip performs: the clip coefficient is computed on the device from the gradient sum of
squares.  Subclasses ``torch.optim.Optimizer`` only so that LR schedulers (CosineAnnealingLR,
finetune_swinir.py:307-309) can drive ``param_groups[0]['lr']``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from ._lib import check, lib
from .network_swinir import SwinIR


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model: SwinIR, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None, grad_div: float = 1.0):
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.max_grad_norm = max_grad_norm
        self.grad_div = float(grad_div)          # world size for data-parallel gradient averaging
        self._step = 0
        self._m: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self._sumsq: Optional[torch.Tensor] = None
        self._ranges: Optional[List[Tuple[int, int]]] = None
        self._engine_id = None

    def _prepare(self):
        eng = self.model._engine
        if eng is None:
            raise RuntimeError("FusedAdamW.step() before the model ran on the GPU")
        if self._engine_id != id(eng):
            if self._m is not None and self._m.numel() == eng.flat.numel():
                self._m, self._v = self._m.to(eng.device), self._v.to(eng.device)
            else:
                self._m, self._v = torch.zeros_like(eng.flat), torch.zeros_like(eng.flat)
            self._sumsq = torch.zeros(1, dtype=torch.float32, device=eng.device)
            trainable = {n for n, p in self.model.named_parameters() if p.requires_grad}
            ranges: List[Tuple[int, int]] = []
            for info in eng.plan.params:          # merge adjacent trainable tensors (64-float aligned slots)
                if info.name not in trainable:
                    continue
                b, e = info.offset, info.offset + (info.numel + 63) // 64 * 64
                if ranges and ranges[-1][1] == b:
                    ranges[-1] = (ranges[-1][0], e)
                else:
                    ranges.append((b, e))
            self._ranges = ranges
            self._engine_id = id(eng)
        return eng

    @torch.no_grad()
    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the (averaged) gradients as a device tensor (no host sync)."""
        eng = self._prepare()
        g = eng.ensure_grad()
        with torch.cuda.device(eng.device):
            self._sumsq.zero_()
            for b, e in self._ranges:
                check(lib().srk_grad_sumsq(g.data_ptr() + 4 * b, e - b, self._sumsq.data_ptr(), _stream(eng.device)))
            return self._sumsq.sqrt() / self.grad_div

    @torch.no_grad()
    def step(self, closure=None, nonfinite: Optional[torch.Tensor] = None):
        """clip + AdamW.  `nonfinite`: optional int32 device counter (training.l1_loss_checked); when it is non-zero -- or
        when the gradient norm is NaN/Inf -- the kernels leave weights and moments untouched (no host sync needed), so a
        bad batch cannot destroy the model before the caller's finite check raises (finetune_swinir.py:159-165)."""
        if closure is not None:
            raise RuntimeError("FusedAdamW does not support closures")
        eng = self._prepare()
        g = eng.ensure_grad()
        grp = self.param_groups[0]
        clip = self.max_grad_norm if self.max_grad_norm and self.max_grad_norm > 0 else 0.0
        bad = None
        if nonfinite is not None:
            if nonfinite.dtype != torch.int32 or nonfinite.device != eng.flat.device:
                raise ValueError("nonfinite must be an int32 tensor on the model's device")
            bad = nonfinite.data_ptr()
        with torch.cuda.device(eng.device):
            st = _stream(eng.device)
            self._sumsq.zero_()          # always computed: a NaN/Inf norm gates the step even without clipping
            for b, e in self._ranges:
                check(lib().srk_grad_sumsq(g.data_ptr() + 4 * b, e - b, self._sumsq.data_ptr(), st))
            self._step += 1
            b1, b2 = grp["betas"]
            for b, e in self._ranges:
                check(lib().srk_adamw_clip_step(eng.flat.data_ptr() + 4 * b, g.data_ptr() + 4 * b, self._m.data_ptr() + 4 * b,
                                                self._v.data_ptr() + 4 * b, e - b, self._sumsq.data_ptr(), float(clip),
                                                self.grad_div, float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                                                float(grp["weight_decay"]), self._step, bad, st))
        eng.packed_valid = False
        return None

    def zero_grad(self, set_to_none: bool = True):
        eng = self.model._engine
        if set_to_none or eng is None or eng.flat_grad is None:
            for p in self.model.parameters():
                p.grad = None
        else:
            eng.flat_grad.zero_()

    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"step": self._step, "exp_avg": self._m, "exp_avg_sq": self._v}
        return sd

    def load_state_dict(self, state_dict):
        fused = state_dict.get("fused")
        rest = {k: v for k, v in state_dict.items() if k != "fused"}
        super().load_state_dict(rest)
        if fused is not None:
            self._step = int(fused["step"])
            self._m, self._v = fused["exp_avg"], fused["exp_avg_sq"]
            self._engine_id = None
