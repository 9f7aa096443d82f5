"""CPU oracle for the HAT path (reference modules/hat_arch.py) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product package
(tpu_superresolution_amd/) never does.  Pinned: oracle/make_golden.py imports the reference's hat_arch.py (timm stand-in,
oracle/ref_import.py) and writes tests/golden/g13_*.npz; tests/test_oracle_golden.py::test_g13_* check this file against them.

A restatement, not a copy: the model is a pure function of a reference-keyed state_dict; index tables come from closed
forms (no meshgrid bookkeeping); the overlapping K/V windows are one gather with a validity mask instead of nn.Unfold +
einops; the shift mask is the arithmetic region label of swinir_oracle.

    calculate_rpi_sa   hat_arch.py:881-894   rpi[p, q] = (yp - yq + ws - 1) (2 ws - 1) + (xp - xq + ws - 1)
    calculate_rpi_oca  hat_arch.py:896-918   rpi[p, k] = (yk - yp + ws - wse + 1) (ws + wse - 1) + (xk - xp + ws - wse + 1),
                       p in the ws x ws window, k in the wse x wse extended window (wse = ws + int(overlap_ratio ws)).
                       It is NEGATIVE for part of its range (-880 .. 640 at ws 16) and the reference indexes the 1521-row
                       table with it as-is: Python / torch wrap negative indices, so row = idx + 1521 when idx < 0.
    calculate_mask     hat_arch.py:921-941   as SwinIR: three h-slices x three w-slices, -100 where labels differ
    HAB.forward        hat_arch.py:281-325   x + attn(LN1 x) + 0.01 CAB(LN1 x);  then  x + mlp(LN2 x)
    CAB                hat_arch.py:41-75     conv3x3(C -> C/3) GELU conv3x3(-> C), times ChannelAttention (avg pool, 1x1 -> C/sq,
                       ReLU, 1x1 -> C, sigmoid)
    OCAB.forward       hat_arch.py:389-439   q: ws x ws windows; k, v: wse x wse windows, stride ws, zero padding (wse-ws)/2
    HAT.forward        hat_arch.py:970-988   only the 'pixelshuffle' head does any work in the reference
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import swinir_oracle as O

Tensor = torch.Tensor


@dataclass(frozen=True)
class HATConfig:
    img_size: int = 64
    patch_size: int = 1
    in_chans: int = 3
    embed_dim: int = 96
    depths: Tuple[int, ...] = (6, 6, 6, 6)
    num_heads: Tuple[int, ...] = (6, 6, 6, 6)
    window_size: int = 7
    compress_ratio: int = 3
    squeeze_factor: int = 30
    conv_scale: float = 0.01
    overlap_ratio: float = 0.5
    mlp_ratio: float = 4.0
    qk_scale: Optional[float] = None
    upscale: int = 2
    img_range: float = 1.0
    upsampler: str = ""
    resi_connection: str = "1conv"

    @staticmethod
    def sr_x4() -> "HATConfig":
        """Official HAT-SRx4 hyper-parameters (BASELINE cfg4; not instantiated anywhere in the reference repo)."""
        return HATConfig(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
                         overlap_ratio=0.5, img_range=1.0, depths=(6,) * 6, embed_dim=180, num_heads=(6,) * 6, mlp_ratio=2.0,
                         upsampler="pixelshuffle", resi_connection="1conv")

    @property
    def wse(self) -> int:
        return self.window_size + int(self.overlap_ratio * self.window_size)

    def kwargs(self) -> dict:
        return dict(img_size=self.img_size, patch_size=self.patch_size, in_chans=self.in_chans, embed_dim=self.embed_dim,
                    depths=list(self.depths), num_heads=list(self.num_heads), window_size=self.window_size,
                    compress_ratio=self.compress_ratio, squeeze_factor=self.squeeze_factor, conv_scale=self.conv_scale,
                    overlap_ratio=self.overlap_ratio, mlp_ratio=self.mlp_ratio, qk_scale=self.qk_scale, upscale=self.upscale,
                    img_range=self.img_range, upsampler=self.upsampler, resi_connection=self.resi_connection)


# ---- index tables ---------------------------------------------------------------------------------------------------
def rpi_sa(ws: int) -> np.ndarray:
    return O.relative_position_index(ws)


def rpi_oca(ws: int, wse: int) -> np.ndarray:
    """[ws*ws, wse*wse] int64, negative entries included (hat_arch.py:896-918)."""
    yp, xp = np.divmod(np.arange(ws * ws), ws)
    yk, xk = np.divmod(np.arange(wse * wse), wse)
    off = ws - wse + 1
    return ((yk[None, :] - yp[:, None] + off) * (ws + wse - 1) + (xk[None, :] - xp[:, None] + off)).astype(np.int64)


def oca_bias(table: Tensor, ws: int, wse: int) -> Tensor:
    """table [(ws+wse-1)^2, nH] -> dense [nH, ws*ws, wse*wse]; negative indices wrap like torch indexing."""
    idx = torch.from_numpy(rpi_oca(ws, wse)).reshape(-1)
    idx = torch.where(idx < 0, idx + table.shape[0], idx)
    return table[idx].reshape(ws * ws, wse * wse, -1).permute(2, 0, 1).contiguous()


def sa_bias(table: Tensor, ws: int) -> Tensor:
    idx = torch.from_numpy(rpi_sa(ws)).reshape(-1)
    return table[idx].reshape(ws * ws, ws * ws, -1).permute(2, 0, 1).contiguous()


def overlap_window_index(H: int, W: int, ws: int, wse: int) -> Tuple[np.ndarray, np.ndarray]:
    """Raster token index of key k of window w for nn.Unfold(wse, stride ws, padding (wse - ws) // 2), and its validity
    (False = zero padding).  -> (idx [nW, wse*wse] int64 (0 where invalid), valid [nW, wse*wse] bool)."""
    pad = (wse - ws) // 2
    nWh, nWw = H // ws, W // ws
    wy, wx = np.divmod(np.arange(nWh * nWw), nWw)
    ky, kx = np.divmod(np.arange(wse * wse), wse)
    y = wy[:, None] * ws - pad + ky[None, :]
    x = wx[:, None] * ws - pad + kx[None, :]
    valid = (y >= 0) & (y < H) & (x >= 0) & (x < W)
    return np.where(valid, y * W + x, 0).astype(np.int64), valid


# ---- blocks -------------------------------------------------------------------------------------------------------------
def _heads(t: Tensor, nH: int) -> Tensor:
    b_, n, c = t.shape
    return t.reshape(b_, n, nH, c // nH).permute(0, 2, 1, 3)


def hab(x: Tensor, x_size: Tuple[int, int], sd: Dict[str, Tensor], pre: str, cfg: HATConfig, nH: int, ws: int, shift: int,
        drop_keep: Optional[Tensor] = None) -> Tensor:
    H, W = x_size
    B, L, C = x.shape
    xn = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    # conv branch on the un-shifted normed features
    img = xn.transpose(1, 2).reshape(B, C, H, W)
    cb = F.conv2d(F.gelu(F.conv2d(img, sd[pre + "conv_block.cab.0.weight"], sd[pre + "conv_block.cab.0.bias"], padding=1)),
                  sd[pre + "conv_block.cab.2.weight"], sd[pre + "conv_block.cab.2.bias"], padding=1)
    y = cb.mean(dim=(2, 3), keepdim=True)
    y = torch.sigmoid(F.conv2d(F.relu(F.conv2d(y, sd[pre + "conv_block.cab.3.attention.1.weight"], sd[pre + "conv_block.cab.3.attention.1.bias"])),
                               sd[pre + "conv_block.cab.3.attention.3.weight"], sd[pre + "conv_block.cab.3.attention.3.bias"]))
    conv_x = (cb * y).flatten(2).transpose(1, 2)
    # (shifted-)window attention: one gather map does roll + partition
    idx = torch.from_numpy(O.window_token_index(H, W, ws, shift))            # [nW, ws*ws]
    xw = xn[:, idx.reshape(-1)].reshape(B * idx.shape[0], ws * ws, C)
    qkv = F.linear(xw, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    q, k, v = (_heads(t, nH) for t in qkv.chunk(3, dim=-1))
    scale = cfg.qk_scale or (C // nH) ** -0.5
    attn = (q * scale) @ k.transpose(-2, -1) + sa_bias(sd[pre + "attn.relative_position_bias_table"], ws)[None]
    if shift > 0:
        mask = torch.from_numpy(O.shift_attn_mask(H, W, ws, shift))         # [nW, N, N]
        attn = (attn.reshape(B, -1, nH, ws * ws, ws * ws) + mask[None, :, None]).reshape(-1, nH, ws * ws, ws * ws)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, ws * ws, C)
    out = F.linear(out, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    attn_x = torch.zeros_like(x)
    attn_x[:, idx.reshape(-1)] = out.reshape(B, -1, C)                        # window_reverse + roll back
    fa = fm = 1.0
    if drop_keep is not None:
        fa, fm = drop_keep[0].reshape(B, 1, 1), drop_keep[1].reshape(B, 1, 1)
    x = x + attn_x * fa + conv_x * cfg.conv_scale
    h = F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5),
                                 sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + h * fm


def ocab(x: Tensor, x_size: Tuple[int, int], sd: Dict[str, Tensor], pre: str, cfg: HATConfig, nH: int, ws: int) -> Tensor:
    H, W = x_size
    B, L, C = x.shape
    wse = ws + int(cfg.overlap_ratio * ws)
    xn = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    q, k, v = F.linear(xn, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).chunk(3, dim=-1)    # raster order [B, L, C] each
    qi = torch.from_numpy(O.window_token_index(H, W, ws, 0))                                   # [nW, ws*ws]
    ki_np, valid_np = overlap_window_index(H, W, ws, wse)
    ki, valid = torch.from_numpy(ki_np), torch.from_numpy(valid_np)
    nW = qi.shape[0]
    qw = q[:, qi.reshape(-1)].reshape(B * nW, ws * ws, C)
    kw = (k[:, ki.reshape(-1)].reshape(B, nW, wse * wse, C) * valid[None, :, :, None]).reshape(B * nW, wse * wse, C)
    vw = (v[:, ki.reshape(-1)].reshape(B, nW, wse * wse, C) * valid[None, :, :, None]).reshape(B * nW, wse * wse, C)
    scale = cfg.qk_scale or (C // nH) ** -0.5
    attn = (_heads(qw, nH) * scale) @ _heads(kw, nH).transpose(-2, -1) + oca_bias(sd[pre + "relative_position_bias_table"], ws, wse)[None]
    out = (attn.softmax(-1) @ _heads(vw, nH)).transpose(1, 2).reshape(B * nW, ws * ws, C)
    merged = torch.zeros_like(x)
    merged[:, qi.reshape(-1)] = out.reshape(B, -1, C)
    x = F.linear(merged, sd[pre + "proj.weight"], sd[pre + "proj.bias"]) + x
    h = F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5),
                                 sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x + h


def forward_features(f: Tensor, sd: Dict[str, Tensor], cfg: HATConfig, drop_keep: Optional[Tensor] = None) -> Tensor:
    B, C, H, W = f.shape
    x = f.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], 1e-5)
    res = cfg.img_size // cfg.patch_size
    blk = 0
    for li, depth in enumerate(cfg.depths):
        y = x
        for bi in range(depth):
            ws, shift = O.effective_window(H, W, cfg.window_size, 0 if bi % 2 == 0 else cfg.window_size // 2, (res, res))
            y = hab(y, (H, W), sd, f"layers.{li}.residual_group.blocks.{bi}.", cfg, cfg.num_heads[li], ws, shift,
                    None if drop_keep is None else drop_keep[blk])
            blk += 1
        y = ocab(y, (H, W), sd, f"layers.{li}.residual_group.overlap_attn.", cfg, cfg.num_heads[li], cfg.window_size)
        y = F.conv2d(y.transpose(1, 2).reshape(B, C, H, W), sd[f"layers.{li}.conv.weight"], sd[f"layers.{li}.conv.bias"], padding=1)
        x = y.flatten(2).transpose(1, 2) + x
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    return x.transpose(1, 2).reshape(B, C, H, W)


def hat_forward(sd: Dict[str, Tensor], cfg: HATConfig, x: Tensor, drop_keep: Optional[Tensor] = None) -> Tensor:
    H, W = x.shape[2:]
    ws = cfg.window_size
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph), mode="reflect")
    mean = torch.tensor([0.4488, 0.4371, 0.4040], dtype=x.dtype).reshape(1, 3, 1, 1) if cfg.in_chans == 3 else torch.zeros(1, 1, 1, 1, dtype=x.dtype)
    x = (x - mean) * cfg.img_range
    s = cfg.upscale
    if cfg.upsampler == "pixelshuffle":
        f = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
        f = F.conv2d(forward_features(f, sd, cfg, drop_keep), sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1) + f
        f = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], padding=1), 0.01)
        if s & (s - 1) == 0:
            for i in range(int(math.log2(s))):
                f = O.pixel_shuffle(F.conv2d(f, sd[f"upsample.{2 * i}.weight"], sd[f"upsample.{2 * i}.bias"], padding=1), 2)
        elif s == 3:
            f = O.pixel_shuffle(F.conv2d(f, sd["upsample.0.weight"], sd["upsample.0.bias"], padding=1), 3)
        else:
            raise ValueError(f"scale {s} is not supported. Supported scales: 2^n and 3.")
        x = F.conv2d(f, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    x = x / cfg.img_range + mean                 # other upsampler strings: the reference returns the re-normalised input
    return x[:, :, :H * s, :W * s]


# ---- state_dict schema + deterministic weights ----------------------------------------------------------------------------
def state_dict_schema(cfg: HATConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) in the reference's state_dict() order (checked against the live reference by make_golden.py)."""
    C, ws, wse = cfg.embed_dim, cfg.window_size, cfg.wse
    hid = int(C * cfg.mlp_ratio)
    out: List[Tuple[str, Tuple[int, ...]]] = [("relative_position_index_SA", (ws * ws, ws * ws)),
                                               ("relative_position_index_OCA", (ws * ws, wse * wse)),
                                               ("conv_first.weight", (C, cfg.in_chans, 3, 3)), ("conv_first.bias", (C,)),
                                               ("patch_embed.norm.weight", (C,)), ("patch_embed.norm.bias", (C,))]

    def lin(name, o, i):
        out.extend([(name + ".weight", (o, i)), (name + ".bias", (o,))])

    def conv(name, o, i, k=3):
        out.extend([(name + ".weight", (o, i, k, k)), (name + ".bias", (o,))])

    def norm(name):
        out.extend([(name + ".weight", (C,)), (name + ".bias", (C,))])

    for li, (depth, nH) in enumerate(zip(cfg.depths, cfg.num_heads)):
        for bi in range(depth):
            p = f"layers.{li}.residual_group.blocks.{bi}."
            norm(p + "norm1")
            out.append((p + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, nH)))
            lin(p + "attn.qkv", 3 * C, C)
            lin(p + "attn.proj", C, C)
            conv(p + "conv_block.cab.0", C // cfg.compress_ratio, C)
            conv(p + "conv_block.cab.2", C, C // cfg.compress_ratio)
            conv(p + "conv_block.cab.3.attention.1", C // cfg.squeeze_factor, C, 1)
            conv(p + "conv_block.cab.3.attention.3", C, C // cfg.squeeze_factor, 1)
            norm(p + "norm2")
            lin(p + "mlp.fc1", hid, C)
            lin(p + "mlp.fc2", C, hid)
        p = f"layers.{li}.residual_group.overlap_attn."
        out.append((p + "relative_position_bias_table", ((ws + wse - 1) ** 2, nH)))
        norm(p + "norm1")
        lin(p + "qkv", 3 * C, C)
        lin(p + "proj", C, C)
        norm(p + "norm2")
        lin(p + "mlp.fc1", hid, C)
        lin(p + "mlp.fc2", C, hid)
        conv(f"layers.{li}.conv", C, C)
    norm("norm")
    conv("conv_after_body", C, C)
    if cfg.upsampler == "pixelshuffle":
        conv("conv_before_upsample.0", 64, C)
        s = cfg.upscale
        if s & (s - 1) == 0:
            for i in range(int(math.log2(s))):
                conv(f"upsample.{2 * i}", 256, 64)
        elif s == 3:
            conv("upsample.0", 576, 64)
        conv("conv_last", cfg.in_chans, 64)
    return out


def random_state_dict(cfg: HATConfig, seed: int = 42, scale: float = 1.0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape in state_dict_schema(cfg):
        if key == "relative_position_index_SA":
            sd[key] = torch.from_numpy(rpi_sa(cfg.window_size))
        elif key == "relative_position_index_OCA":
            sd[key] = torch.from_numpy(rpi_oca(cfg.window_size, cfg.wse))
        elif "norm" in key and key.endswith("weight"):
            sd[key] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif "norm" in key and key.endswith("bias"):
            sd[key] = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 4:
            b = scale / math.sqrt(shape[1] * shape[2] * shape[3])
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * b
        elif key.endswith("bias"):
            sd[key] = 0.02 * torch.randn(shape, generator=g)
        elif key.endswith("relative_position_bias_table"):
            sd[key] = 0.2 * torch.randn(shape, generator=g)          # larger than the init's 0.02 so a wrong index shows up
        else:
            sd[key] = 0.02 * scale * torch.randn(shape, generator=g)
    return sd


def param_keys(cfg: HATConfig) -> List[str]:
    return [k for k, _ in state_dict_schema(cfg) if not k.startswith("relative_position_index")]
