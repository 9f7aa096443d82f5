"""Host-side driver of the libsrk SwinIR executor: plan, flat parameter/gradient buffers, packing,
workspace and the autograd bridge.  PyTorch is used for device memory, streams and autograd
plumbing only; all arithmetic happens in libsrk.so.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SwinIRConfig, check, lib


def _stream(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class ParamInfo:
    __slots__ = ("name", "offset", "numel", "shape")

    def __init__(self, name: str, offset: int, numel: int, shape: Tuple[int, ...]):
        self.name, self.offset, self.numel, self.shape = name, offset, numel, shape


class SwinIRPlan:
    """Owns a srk_swinir_plan (host bookkeeping only)."""

    def set_option(self, name: str, value: int) -> None:
        """A kernel option of THIS plan (names as srk_set_option): applies to its later calls only, whatever the thread's defaults."""
        check(lib().srk_swinir_plan_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str):
        """-> (value, carried by the plan?)"""
        v, is_set = C.c_int(), C.c_int()
        check(lib().srk_swinir_plan_get_option(self.handle, name.encode(), C.byref(v), C.byref(is_set)))
        return v.value, bool(is_set.value)

    def __init__(self, *, img_size: int, in_chans: int, embed_dim: int, depths: Sequence[int], num_heads: Sequence[int],
                 window_size: int, mlp_ratio: float, upscale: int, img_range: float, upsampler: str,
                 qk_scale: Optional[float] = None, resi_connection: str = "1conv", use_checkpoint: bool = False,
                 ape: bool = False, options: Optional[dict] = None):
        # any other upsampler string takes the reference's `else` branch: the denoising head (network_swinir.py:760-762)
        ups = _lib.UPSAMPLERS.get(upsampler, _lib.UPSAMPLER_NONE)
        if resi_connection not in _lib.RESI:
            raise ValueError(f"unknown resi_connection {resi_connection!r}")
        if len(depths) != len(num_heads) or len(depths) > 16:
            raise ValueError("depths / num_heads must have the same length (<= 16)")
        cfg = SwinIRConfig()
        cfg.img_size = int(img_size)
        cfg.in_chans = int(in_chans)
        cfg.embed_dim = int(embed_dim)
        cfg.num_layers = len(depths)
        for i, (d, h) in enumerate(zip(depths, num_heads)):
            cfg.depths[i] = int(d)
            cfg.num_heads[i] = int(h)
        cfg.window_size = int(window_size)
        cfg.hidden_dim = int(embed_dim * mlp_ratio)
        cfg.upscale = int(upscale)
        cfg.upsampler = ups
        cfg.img_range = float(img_range)
        mean = (0.4488, 0.4371, 0.4040) if in_chans == 3 else (0.0, 0.0, 0.0)   # network_swinir.py:658-662
        for i in range(3):
            cfg.mean[i] = mean[i]
        cfg.qk_scale = float(qk_scale) if qk_scale else 0.0
        cfg.resi_connection = _lib.RESI[resi_connection]
        cfg.use_checkpoint = 1 if use_checkpoint else 0
        cfg.ape = 1 if ape else 0
        self.cfg = cfg
        self.upscale = int(upscale)
        self.n_blocks = int(sum(depths))
        handle = C.c_void_p()
        check(lib().srk_swinir_plan_create(C.byref(cfg), C.byref(handle)))
        self.handle = handle
        self.param_floats = int(lib().srk_swinir_param_floats(handle))
        self.params: List[ParamInfo] = []
        name = C.c_char_p()
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(lib().srk_swinir_param_count(handle)):
            check(lib().srk_swinir_param_info(handle, i, C.byref(name), C.byref(off), C.byref(numel), C.byref(ndim), C.byref(shape)))
            self.params.append(ParamInfo(name.value.decode(), off.value, numel.value, tuple(shape[j] for j in range(ndim.value))))
        self.num_segments = int(lib().srk_swinir_num_segments(handle))
        for k, v in (options or {}).items():       # per-plan kernel options (include/srk.h: srk_swinir_plan_set_option)
            self.set_option(k, v)
        self.segment_ranges: List[Tuple[int, int]] = []
        b, e = C.c_int64(), C.c_int64()
        for s in range(self.num_segments):
            check(lib().srk_swinir_segment_range(handle, s, C.byref(b), C.byref(e)))
            self.segment_ranges.append((b.value, e.value))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib().srk_swinir_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class SwinIREngine:
    """Device state for one model replica on one GPU."""

    def __init__(self, plan: SwinIRPlan, device: torch.device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("SwinIR HIP path needs a GPU device; there is no CPU fallback in this package")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        _lib.claim_device(device.index)
        self.plan = plan
        self.device = device
        self.flat = torch.zeros(plan.param_floats, dtype=torch.float32, device=device)
        self.flat_grad: Optional[torch.Tensor] = None
        self.const = torch.empty(max(1, lib().srk_swinir_const_bytes(plan.handle)), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            check(lib().srk_swinir_const_init(plan.handle, self.const.data_ptr(), _stream(device)))
        self.packed = torch.empty(lib().srk_swinir_packed_bytes(plan.handle), dtype=torch.uint8, device=device)
        self.workspace: Optional[torch.Tensor] = None
        self._ws_key = None
        self.packed_valid = False
        self.pack_version = -1            # sum of the parameters' autograd version counters at the last pack
        self.generation = 0               # bumped by every training forward: ties an autograd node to "its" workspace
        self.segment_hook: Optional[Callable[[int, int, int], None]] = None

    # -- parameters -------------------------------------------------------------------------------
    def views(self, base: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {p.name: base[p.offset:p.offset + p.numel].view(p.shape) for p in self.plan.params}

    def pack(self) -> None:
        with torch.cuda.device(self.device):
            check(lib().srk_swinir_pack(self.plan.handle, self.flat.data_ptr(), self.packed.data_ptr(), _stream(self.device)))
        self.packed_valid = True

    def ensure_grad(self) -> torch.Tensor:
        if self.flat_grad is None:
            self.flat_grad = torch.zeros_like(self.flat)
        return self.flat_grad

    # -- execution --------------------------------------------------------------------------------
    def _workspace(self, B: int, H: int, W: int, training: bool) -> torch.Tensor:
        key = (B, H, W, bool(training))
        need = lib().srk_swinir_workspace_bytes(self.plan.handle, B, H, W, int(training))
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = None
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ws_key = key
        return self.workspace

    def forward(self, x: torch.Tensor, training: bool, drop_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("SwinIR HIP path needs a GPU tensor; there is no CPU fallback in this package")
        if x.device != self.device:
            raise RuntimeError(f"input is on {x.device} but the model is bound to {self.device}")
        x = x.contiguous().float()
        B, Cin, H, W = x.shape
        if Cin != self.plan.cfg.in_chans:
            raise ValueError(f"expected {self.plan.cfg.in_chans} input channels, got {Cin}")
        if not self.packed_valid:
            self.pack()
        ws = self._workspace(B, H, W, training)
        s = self.plan.upscale
        y = torch.empty((B, Cin, H * s, W * s), dtype=torch.float32, device=self.device)
        ds = drop_scale.contiguous().data_ptr() if drop_scale is not None else None
        if training:
            self.generation += 1
        with torch.cuda.device(self.device):      # kernels launch on the current device: make it the tensors' device
            check(lib().srk_swinir_forward(self.plan.handle, self.flat.data_ptr(), self.packed.data_ptr(), x.data_ptr(),
                                           y.data_ptr(), ws.data_ptr(), B, H, W, int(training), ds, _stream(self.device)))
        return y

    def forward_features(self, f: torch.Tensor) -> torch.Tensor:
        """SwinIR.forward_features (network_swinir.py:790-803): conv_first output [B,C,H,W] -> [B,C,H,W]; inference only."""
        if not f.is_cuda or f.device != self.device:
            raise RuntimeError(f"forward_features needs a tensor on {self.device} (no CPU fallback in this package)")
        f = f.contiguous().float()
        B, Cc, H, W = f.shape
        if Cc != self.plan.cfg.embed_dim:
            raise ValueError(f"expected {self.plan.cfg.embed_dim} feature channels, got {Cc}")
        if not self.packed_valid:
            self.pack()
        ws = self._workspace(B, H, W, False)
        out = torch.empty_like(f)
        with torch.cuda.device(self.device):
            check(lib().srk_swinir_forward_features(self.plan.handle, self.flat.data_ptr(), self.packed.data_ptr(), f.data_ptr(),
                                                    out.data_ptr(), ws.data_ptr(), B, H, W, _stream(self.device)))
        return out

    def backward(self, d_y: torch.Tensor, shape: Tuple[int, int, int], drop_scale: Optional[torch.Tensor] = None) -> None:
        """Accumulates parameter gradients into flat_grad; calls segment_hook(seg, begin, end) after each segment."""
        B, H, W = shape
        if self._ws_key != (B, H, W, True):
            raise RuntimeError("backward without a matching training forward")
        g = self.ensure_grad()
        d_y = d_y.contiguous().float()
        ds = drop_scale.contiguous().data_ptr() if drop_scale is not None else None
        with torch.cuda.device(self.device):
            for seg in range(self.plan.num_segments):
                check(lib().srk_swinir_backward(self.plan.handle, self.flat.data_ptr(), self.packed.data_ptr(), g.data_ptr(),
                                                d_y.data_ptr(), self.workspace.data_ptr(), B, H, W, ds, seg, seg + 1,
                                                _stream(self.device)))
                if self.segment_hook is not None:
                    b, e = self.plan.segment_ranges[seg]
                    self.segment_hook(seg, b, e)

    def activation(self, name: str, dtype: torch.dtype) -> torch.Tensor:
        """Debug/parity view of a named workspace buffer (flat)."""
        off, nbytes = C.c_size_t(), C.c_size_t()
        check(lib().srk_swinir_workspace_lookup(self.plan.handle, name.encode(), C.byref(off), C.byref(nbytes)))
        return self.workspace[off.value:off.value + nbytes.value].view(dtype)


class _SwinIRFunction(torch.autograd.Function):
    """Autograd bridge: one node for the whole model.  Parameter gradients are written by the C backward
    straight into the engine's flat gradient buffer (p.grad are views of it), not returned to autograd."""

    @staticmethod
    def forward(ctx, x, anchor, module, drop_scale, training):
        eng: SwinIREngine = module._engine
        y = eng.forward(x, training, drop_scale)
        ctx.module = module
        ctx.shape = (x.shape[0], x.shape[2], x.shape[3])
        ctx.drop_scale = drop_scale
        ctx.trained = training
        ctx.generation = eng.generation
        return y

    @staticmethod
    def backward(ctx, d_y):
        module = ctx.module
        if not ctx.trained:
            raise RuntimeError("SwinIR forward was run without gradient tracking")
        eng = module._engine
        if eng is None or ctx.generation != eng.generation:
            # one activation workspace per engine: a later grad-enabled forward has overwritten what this node saved
            raise RuntimeError("SwinIR.backward: the saved activations of this forward were overwritten by a later "
                               "grad-enabled forward of the same model; call backward() before the next training forward "
                               "(one live autograd graph per model)")
        module._backward_into_flat(d_y, ctx.shape, ctx.drop_scale)
        return None, None, None, None, None
