"""Thin torch wrappers over the building-block entry points of libsrk.so.

Tensors must live on the GPU ("cuda" == ROCm/HIP device in PyTorch-ROCm) and be contiguous; every
call is enqueued on the current torch stream.  Nothing here computes on the CPU: a CPU tensor or a
missing library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import threading

import torch

from . import _lib
from ._lib import WinGeom, check, lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libsrk operates on GPU tensors only (got a CPU tensor); there is no CPU fallback")
    if t.device.index != torch.cuda.current_device():
        # the kernel would launch on the current device with another device's pointers (a GPU fault, not a Python error)
        raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "libsrk ops launch on the current device (torch.cuda.set_device / one process per GPU)")
    if not t.is_contiguous():
        raise RuntimeError("libsrk needs contiguous tensors")
    return t.data_ptr()


def _geom(H: int, W: int, shift: int):
    return C.byref(WinGeom(H, W, shift))


# ---------------------------------------------------------------------------------------------------
# bit-exact index ops
# ---------------------------------------------------------------------------------------------------
def window_partition(x: torch.Tensor, window_size: int) -> torch.Tensor:
    """reference window_partition (network_swinir.py:33-45): (B,H,W,C) -> (B*nW, ws, ws, C)."""
    B, H, W, Cc = x.shape
    x = x.contiguous()
    out = torch.empty((B * (H // window_size) * (W // window_size), window_size, window_size, Cc), dtype=x.dtype, device=x.device)
    check(lib().srk_window_partition(_p(x), _p(out), B, H, W, Cc, window_size, x.element_size(), _stream()))
    return out


def window_reverse(windows: torch.Tensor, window_size: int, H: int, W: int) -> torch.Tensor:
    """reference window_reverse (network_swinir.py:48-62): (B*nW, ws, ws, C) -> (B,H,W,C)."""
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    Cc = windows.shape[-1]
    windows = windows.contiguous()
    out = torch.empty((B, H, W, Cc), dtype=windows.dtype, device=windows.device)
    check(lib().srk_window_reverse(_p(windows), _p(out), B, H, W, Cc, window_size, windows.element_size(), _stream()))
    return out


def roll2d(x: torch.Tensor, shifts: Tuple[int, int]) -> torch.Tensor:
    """torch.roll(x, shifts, dims=(1, 2)) on (B,H,W,C)."""
    B, H, W, Cc = x.shape
    x = x.contiguous()
    out = torch.empty_like(x)
    check(lib().srk_roll2d(_p(x), _p(out), B, H, W, Cc, int(shifts[0]), int(shifts[1]), x.element_size(), _stream()))
    return out


def pixel_shuffle(x: torch.Tensor, r: int) -> torch.Tensor:
    B, Crr, H, W = x.shape
    Cc = Crr // (r * r)
    x = x.contiguous()
    out = torch.empty((B, Cc, H * r, W * r), dtype=x.dtype, device=x.device)
    check(lib().srk_pixel_shuffle(_p(x), _p(out), B, Cc, H, W, r, x.element_size(), _stream()))
    return out


def shift_mask(H: int, W: int, window_size: int, shift: int, device="cuda") -> torch.Tensor:
    N = window_size * window_size
    out = torch.empty(((H // window_size) * (W // window_size), N, N), dtype=torch.float32, device=device)
    check(lib().srk_shift_mask(_p(out), H, W, window_size, shift, _stream()))
    return out


def relative_position_index(window_size: int, device="cuda") -> torch.Tensor:
    N = window_size * window_size
    out = torch.empty((N, N), dtype=torch.int64, device=device)
    check(lib().srk_relative_position_index(_p(out), window_size, _stream()))
    return out


# ---------------------------------------------------------------------------------------------------
# building blocks in the kernels' internal (padded) layouts
# ---------------------------------------------------------------------------------------------------
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, C_real: int, *, geom=None,
                  out_bf16: bool = True, out_f32: bool = False):
    """x fp32 [rows, CP] -> (y_bf16 | None, y_f32 | None, mean, rstd).  geom=(H, W, shift) -> window order."""
    rows, CP = x.shape
    yb = torch.empty((rows, CP), dtype=torch.bfloat16, device=x.device) if out_bf16 else None
    yf = torch.empty((rows, CP), dtype=torch.float32, device=x.device) if out_f32 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    g = _geom(*geom) if geom is not None else None
    check(lib().srk_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(yb), _p(yf), _p(mean), _p(rstd), rows, C_real, CP, g, _stream()))
    return yb, yf, mean, rstd


def window_attention_fwd(qkv: torch.Tensor, bias_dense: torch.Tensor, H: int, W: int, shift: int) -> torch.Tensor:
    """qkv bf16 [3, B_, nH, 64, 32]; bias_dense fp32 [nH, 64, 64] -> out bf16 [B_*64, nH*32]."""
    _, B_, nH, N, D = qkv.shape
    assert N == 64 and D == 32
    out = torch.empty((B_ * 64, nH * 32), dtype=torch.bfloat16, device=qkv.device)
    check(lib().srk_window_attention_fwd(_p(qkv), _p(bias_dense), _p(out), B_, nH, _geom(H, W, shift), _stream()))
    return out


def window_attention_bwd(qkv: torch.Tensor, bias_dense: torch.Tensor, d_out: torch.Tensor, scale: float, H: int, W: int,
                         shift: int):
    """-> (d_qkv bf16 [B_*64, 3*nH*32], d_table fp32 [225, nH])."""
    _, B_, nH, _, _ = qkv.shape
    d_qkv = torch.empty((B_ * 64, 3 * nH * 32), dtype=torch.bfloat16, device=qkv.device)
    d_table = torch.zeros((225, nH), dtype=torch.float32, device=qkv.device)
    slab = torch.empty(lib().srk_window_attention_bwd_scratch(B_, nH), dtype=torch.uint8, device=qkv.device)
    check(lib().srk_window_attention_bwd(_p(qkv), _p(bias_dense), _p(d_out), _p(d_qkv), _p(d_table), _p(slab), B_, nH,
                                         float(scale), _geom(H, W, shift), _stream()))
    return d_qkv, d_table


def window_attention_bwd_fused(xn: torch.Tensor, w_qkv: torch.Tensor, b_qkv: Optional[torch.Tensor], scale: float, d_x1: torch.Tensor,
                               w_proj_t: torch.Tensor, bias_dense: torch.Tensor, H: int, W: int, shift: int):
    """Attention backward with q/k/v re-projected from xn and the output-projection dgrad folded in (classical width).
    xn, d_x1 bf16 [B_*64, 192] window order; w_qkv bf16 [576, 192]; w_proj_t bf16 [192, 192]
    -> (d_qkv bf16 [B_*64, 576], d_table fp32 [225, 6])."""
    B_ = xn.shape[0] // 64
    nH = bias_dense.shape[0]
    d_qkv = torch.empty((B_ * 64, 3 * nH * 32), dtype=torch.bfloat16, device=xn.device)
    d_table = torch.zeros((225, nH), dtype=torch.float32, device=xn.device)
    slab = torch.empty(max(16, lib().srk_window_attention_bwd_fused_scratch(B_, nH)), dtype=torch.uint8, device=xn.device)
    check(lib().srk_window_attention_bwd_fused(_p(xn), xn.stride(0), _p(w_qkv), _p(b_qkv), float(scale), _p(d_x1), d_x1.stride(0),
                                               _p(w_proj_t), _p(bias_dense), _p(d_qkv), _p(d_table), _p(slab), B_, nH,
                                               _geom(H, W, shift), _stream()))
    return d_qkv, d_table


def rel_pos_bias_expand(table: torch.Tensor) -> torch.Tensor:
    nH = table.shape[1]
    out = torch.empty((nH, 64, 64), dtype=torch.float32, device=table.device)
    check(lib().srk_rel_pos_bias_expand(_p(table), _p(out), nH, _stream()))
    return out


def linear_bf16(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    M, K = a.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    check(lib().srk_linear_bf16(_p(a), _p(w), _p(bias), _p(y), M, N, K, _stream()))
    return y


_WGRAD_WS = {}


def _bind_wgrad_workspace(device: torch.device) -> None:
    """Register this thread's weight-gradient workspace (a cached torch tensor: the caller owns the memory, the library never
    allocates) so that the stand-alone wgrad entry points reduce their row-splits in a fixed order."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(int(lib().srk_wgrad_workspace_bytes()), dtype=torch.uint8, device=device)
    check(lib().srk_set_wgrad_workspace(_p(ws), ws.numel()))


def linear_wgrad_bf16(y: torch.Tensor, x: torch.Tensor, with_bias: bool = True):
    M, N = y.shape
    K = x.shape[1]
    dw = torch.zeros((N, K), dtype=torch.float32, device=y.device)
    db = torch.zeros((N,), dtype=torch.float32, device=y.device) if with_bias else None
    _bind_wgrad_workspace(y.device)
    check(lib().srk_linear_wgrad_bf16(_p(y), _p(x), _p(dw), _p(db), M, N, K, _stream()))
    return dw, db


class ZeroArena:
    """Zero-initialised fp32 accumulators of one backward pass carved out of ONE buffer that a single fill clears.  A HAT / DAT backward
    needs ~20 small zeroed tensors per block (weight / bias / LayerNorm gradient accumulators): as separate torch.zeros they are ~750
    fill launches of ~2.7 us each per step.  The arena learns its size in the first pass (requests beyond the buffer fall back to
    torch.zeros) and serves the following passes from one allocation; every piece starts on a 256-byte boundary."""

    def __init__(self):
        self.need = 0            # floats requested in the last pass
        self.buf = None
        self.off = 0

    def begin(self, device) -> None:
        want = max(self.need, self.off)
        self.need = want
        self.buf = torch.zeros(want, dtype=torch.float32, device=device) if want > 0 else None     # a fresh buffer per pass: the previous
        self.off = 0                                                                              # pass's gradients may still be referenced

    def zeros(self, shape, device) -> torch.Tensor:
        shape = tuple(shape) if not isinstance(shape, int) else (shape,)
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        o = self.off
        self.off += n_al
        if self.buf is not None and self.buf.device == device and o + n_al <= self.buf.numel():
            return self.buf[o:o + n].view(shape)
        return torch.zeros(shape, dtype=torch.float32, device=device)


_ARENA = threading.local()


def zeros_f32(shape, device) -> torch.Tensor:
    """fp32 zeros from the backward pass's arena when one is active on this thread (arena_scope), else torch.zeros"""
    ar = getattr(_ARENA, "cur", None)
    if ar is None:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return ar.zeros(shape, device)


class arena_scope:
    """with arena_scope(arena, device): ...   -- zeros_f32 inside the block come from `arena` (not re-entrant across threads)"""

    def __init__(self, arena: "ZeroArena", device):
        self.arena, self.device = arena, device

    def __enter__(self):
        self.prev = getattr(_ARENA, "cur", None)
        self.arena.begin(self.device)
        _ARENA.cur = self.arena
        return self.arena

    def __exit__(self, *exc):
        _ARENA.cur = self.prev
        self.arena.need = max(self.arena.need, self.arena.off)
        return False


def linear_wgrad_multi_bf16(pairs):
    """pairs: up to four (y [M][N] bf16, x [M][K] bf16) with the same M (row-major, contiguous or column slices of contiguous rows) ->
    [(dw [N][K], db [N])] from ONE launch (include/srk.h: srk_linear_wgrad_multi_bf16)."""
    from ._lib import WgradProblem
    assert 1 <= len(pairs) <= 4
    M = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    arr = (WgradProblem * len(pairs))()
    outs = []
    for i, (y, x) in enumerate(pairs):
        assert y.shape[0] == M and x.shape[0] == M and y.stride(1) == 1 and x.stride(1) == 1
        N, K = y.shape[1], x.shape[1]
        dw = zeros_f32((N, K), dev)
        db = zeros_f32((N,), dev)
        arr[i].y, arr[i].ldy, arr[i].x, arr[i].ldx = y.data_ptr(), y.stride(0), x.data_ptr(), x.stride(0)
        arr[i].dw, arr[i].db, arr[i].N, arr[i].K = dw.data_ptr(), db.data_ptr(), N, K
        outs.append((dw, db))
    _bind_wgrad_workspace(dev)
    check(lib().srk_linear_wgrad_multi_bf16(arr, len(pairs), M, _stream()))
    return outs


def conv3x3_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x bf16 NHWC [B,H,W,CinP], w bf16 [N, 9*CinP] (tap-major) -> y bf16 NHWC [B,H,W,N]."""
    B, H, W, CinP = x.shape
    N = w.shape[0]
    y = torch.empty((B, H, W, N), dtype=torch.bfloat16, device=x.device)
    check(lib().srk_conv3x3_bf16(_p(x), _p(w), _p(bias), _p(y), B, H, W, CinP, N, _stream()))
    return y


def conv3x3_wgrad_bf16(dy: torch.Tensor, x: torch.Tensor):
    B, H, W, N = dy.shape
    CinP = x.shape[-1]
    dw = torch.zeros((N, 9 * CinP), dtype=torch.float32, device=x.device)
    db = torch.zeros((N,), dtype=torch.float32, device=x.device)
    _bind_wgrad_workspace(x.device)
    check(lib().srk_conv3x3_wgrad_bf16(_p(dy), _p(x), _p(dw), _p(db), B, H, W, CinP, N, _stream()))
    return dw, db


def probe_trread(tile: torch.Tensor) -> torch.Tensor:
    """tile int16 [64,16] -> fragments int16 [64 lanes, 8]; expected[l, j] = tile[8*(l>>4)+j, l&15]."""
    out = torch.empty((64, 8), dtype=torch.int16, device=tile.device)
    check(lib().srk_probe_trread(_p(tile), _p(out), _stream()))
    return out


def batch_psnr(pred: torch.Tensor, target: torch.Tensor, max_val: float = 1.0, psnr_sum: Optional[torch.Tensor] = None,
               abs_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-image PSNR [B] of fp32 [B, ...] images (finetune_swinir.py:69-74) in one fused pass; optionally ACCUMULATES the
    batch's PSNR sum and sum |pred - target| into the given fp32 scalars (validation loop without per-batch host syncs)."""
    assert pred.shape == target.shape and pred.dtype == torch.float32 and target.dtype == torch.float32
    pred, target = pred.contiguous(), target.contiguous()
    B = pred.shape[0]
    per_image = pred.numel() // B
    ws = torch.empty(int(lib().srk_batch_psnr_workspace(per_image, B)), dtype=torch.uint8, device=pred.device)
    out = torch.empty(B, dtype=torch.float32, device=pred.device)
    check(lib().srk_batch_psnr(_p(pred), _p(target), _p(ws), B, per_image, float(max_val), _p(out), _p(psnr_sum), _p(abs_sum), _stream()))
    return out


def eval_psnr(x: torch.Tensor, y: torch.Tensor, max_val: float = 1.0):
    """evaluate.py:24-29 on the device -> (per-image PSNR [B], batch mean [1]) without a host sync."""
    assert x.shape == y.shape and x.dtype == torch.float32 and y.dtype == torch.float32
    x, y = x.contiguous(), y.contiguous()
    B = x.shape[0]
    per_image = x.numel() // B
    ws = torch.empty(int(lib().srk_eval_psnr_workspace(per_image, B)), dtype=torch.uint8, device=x.device)
    per, mean = torch.empty(B, dtype=torch.float32, device=x.device), torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib().srk_eval_psnr(_p(x), _p(y), _p(ws), B, per_image, float(max_val), _p(per), _p(mean), _stream()))
    return per, mean


def ssim(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0):
    """pytorch_msssim.ssim semantics on the device (csrc/metrics.hip; restated, parity unpinned) -> (per-image [B], batch mean [1])."""
    assert x.shape == y.shape and x.ndim == 4 and x.dtype == torch.float32 and y.dtype == torch.float32
    x, y = x.contiguous(), y.contiguous()
    B, Cc, H, W = x.shape
    ws = torch.empty(max(4, int(lib().srk_ssim_workspace(B, Cc, H, W))), dtype=torch.uint8, device=x.device)
    per, mean = torch.empty(B, dtype=torch.float32, device=x.device), torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib().srk_ssim(_p(x), _p(y), _p(ws), B, Cc, H, W, float(data_range), _p(per), _p(mean), _stream()))
    return per, mean


def l1_loss_fwd_bwd(pred: torch.Tensor, target: torch.Tensor, want_grad: bool = True, grad_scale: float = 1.0):
    """-> (loss fp32 [1], d_pred | None, nonfinite int32 [1])   (finetune_swinir.py:66-67, :133-143)."""
    loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    bad = torch.zeros(1, dtype=torch.int32, device=pred.device)
    d = torch.empty_like(pred) if want_grad else None
    check(lib().srk_l1_loss_fwd_bwd(_p(pred), _p(target), _p(d), _p(loss), _p(bad), pred.numel(), float(grad_scale), _stream()))
    return loss, d, bad
