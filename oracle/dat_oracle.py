"""CPU oracle for the DAT path (reference modules/dat_arch.py) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product package never does.
Pinned: oracle/make_golden.py imports the reference's dat_arch.py (timm stand-in; einops is installed) and writes
tests/golden/g14_*.npz; tests/test_oracle_golden.py::test_g14_* check this file against them.

A restatement, not a copy: the model is a pure function of a reference-keyed state_dict; rectangular window partition /
reverse with the per-branch roll is ONE gather map; masks are arithmetic region labels; the dynamic position bias is
evaluated once per block from the closed-form offset table; BatchNorm is its inference affine (running statistics).

    img2windows / windows2img   dat_arch.py:15-35   rectangular H_sp x W_sp windows (row-major tokens inside a window)
    DynamicPosBias              dat_arch.py:93-130  residual=False: pos3(pos2(pos1(pos_proj(b)))), each pos_k = LN, ReLU, Linear
    Spatial_Attention           dat_arch.py:133-244 window MSA over dim/2 channels with heads/2 heads, bias = pos(rpe_biases)[rpi]
    Adaptive_Spatial_Attention  dat_arch.py:247-438 two branches (H_sp x W_sp and W_sp x H_sp), shifted on blocks
                                (rg even: b = 2, 6, ..; rg odd: b = 0, 4, ..) :297,:391 by (s0, s1) / (s1, s0); DW-conv branch on v;
                                channel / spatial interaction gates; proj
    Adaptive_Channel_Attention  dat_arch.py:441-528 L2-normalised q, k over tokens, (C/h x C/h) attention with temperature
    SGFN                        dat_arch.py:38-90   fc1, GELU, split, x1 * DWconv(LN(x2)), fc2
    DATB / ResidualGroup / DAT  dat_arch.py:531-860

Eval semantics by default (BatchNorm uses running statistics).  ``train_mode(record)`` switches every BatchNorm to batch statistics
(nn.BatchNorm2d in training, momentum 0.1: biased variance for the normalisation, unbiased for the running estimate) and records the
updated running buffers; ``loss_and_grads`` is torch autograd over this functional model -- the training oracle pinned by
tests/golden/g14c_dat_train.npz (the reference's own DAT in .train(), oracle/make_golden.py::gen_g14c).
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
class DATConfig:
    img_size: int = 64
    in_chans: int = 3
    embed_dim: int = 180
    split_size: Tuple[int, int] = (2, 4)
    depth: Tuple[int, ...] = (2, 2, 2, 2)
    num_heads: Tuple[int, ...] = (2, 2, 2, 2)
    expansion_factor: float = 4.0
    qkv_bias: bool = True
    qk_scale: Optional[float] = None
    upscale: int = 2
    img_range: float = 1.0
    resi_connection: str = "1conv"
    upsampler: str = "pixelshuffle"

    @staticmethod
    def sr_x4() -> "DATConfig":
        """Official DAT x4 hyper-parameters (BASELINE cfg5; the reference repo only has a DAT-S-like __main__ demo)."""
        return DATConfig(upscale=4, in_chans=3, img_size=64, img_range=1.0, depth=(6,) * 6, embed_dim=180, num_heads=(6,) * 6,
                         expansion_factor=4.0, resi_connection="1conv", split_size=(8, 32), upsampler="pixelshuffle")

    def kwargs(self) -> dict:
        return dict(img_size=self.img_size, in_chans=self.in_chans, embed_dim=self.embed_dim, split_size=list(self.split_size),
                    depth=list(self.depth), num_heads=list(self.num_heads), expansion_factor=self.expansion_factor,
                    qkv_bias=self.qkv_bias, qk_scale=self.qk_scale, upscale=self.upscale, img_range=self.img_range,
                    resi_connection=self.resi_connection, upsampler=self.upsampler)


def is_shifted(rg: int, b: int) -> bool:
    """dat_arch.py:297 / :391"""
    return (rg % 2 == 0 and b > 0 and (b - 2) % 4 == 0) or (rg % 2 != 0 and b % 4 == 0)


# ---- index tables ---------------------------------------------------------------------------------------------------
def rect_window_token_index(H: int, W: int, hs: int, wsz: int, sy: int, sx: int) -> np.ndarray:
    """[nW, hs*wsz] raster token index of window-order row p of window w after roll(-sy, -sx) + img2windows(hs, wsz)."""
    nWh, nWw = H // hs, W // wsz
    wy, wx = np.divmod(np.arange(nWh * nWw), nWw)
    py, px = np.divmod(np.arange(hs * wsz), wsz)
    y = (wy[:, None] * hs + py[None, :] + sy) % H
    x = (wx[:, None] * wsz + px[None, :] + sx) % W
    return (y * W + x).astype(np.int64)


def rect_shift_mask(H: int, W: int, hs: int, wsz: int, sy: int, sx: int) -> np.ndarray:
    """[nW, N, N] in {0, -100}: region labels from the slices (0, -win), (-win, -shift), (-shift, end) per axis (:334-380)."""
    def lab(n, w, s):
        v = np.arange(n)
        return np.where(v < n - w, 0, np.where(v < n - s, 1, 2))
    label = lab(H, hs, sy)[:, None] * 3 + lab(W, wsz, sx)[None, :]
    win = label.reshape(H // hs, hs, W // wsz, wsz).transpose(0, 2, 1, 3).reshape(-1, hs * wsz)
    return np.where(win[:, None, :] != win[:, :, None], -100.0, 0.0).astype(np.float32)


def rpe_offsets(hs: int, wsz: int) -> np.ndarray:
    """rpe_biases [(2hs-1)(2wsz-1), 2] float32 (:176-180)."""
    dy, dx = np.meshgrid(np.arange(1 - hs, hs), np.arange(1 - wsz, wsz), indexing="ij")
    return np.stack([dy.reshape(-1), dx.reshape(-1)], 1).astype(np.float32)


def rect_rpi(hs: int, wsz: int) -> np.ndarray:
    """relative_position_index [N, N] (:183-193)."""
    py, px = np.divmod(np.arange(hs * wsz), wsz)
    return ((py[:, None] - py[None, :] + hs - 1) * (2 * wsz - 1) + (px[:, None] - px[None, :] + wsz - 1)).astype(np.int64)


def dynamic_pos_bias(sd: Dict[str, Tensor], pre: str, hs: int, wsz: int) -> Tensor:
    """-> dense [heads, N, N] (:93-130 with residual=False, gathered as :219-224)."""
    p = F.linear(torch.from_numpy(rpe_offsets(hs, wsz)), sd[pre + "pos_proj.weight"], sd[pre + "pos_proj.bias"])
    for k in ("pos1", "pos2", "pos3"):
        p = F.layer_norm(p, (p.shape[-1],), sd[f"{pre}{k}.0.weight"], sd[f"{pre}{k}.0.bias"], 1e-5)
        p = F.linear(F.relu(p), sd[f"{pre}{k}.2.weight"], sd[f"{pre}{k}.2.bias"])
    N = hs * wsz
    return p[torch.from_numpy(rect_rpi(hs, wsz)).reshape(-1)].reshape(N, N, -1).permute(2, 0, 1).contiguous()


# ---- pieces ---------------------------------------------------------------------------------------------------------------
_TRAIN: Optional[dict] = None      # None: eval; a dict: train mode, filled with the updated running statistics


class train_mode:
    """with train_mode(record): BatchNorm uses batch statistics; record[key] = new running_mean / running_var / num_batches_tracked."""

    def __init__(self, record: dict):
        self.record = record

    def __enter__(self):
        global _TRAIN
        self.prev, _TRAIN = _TRAIN, self.record
        return self.record

    def __exit__(self, *exc):
        global _TRAIN
        _TRAIN = self.prev


def bn_eval(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """BatchNorm2d on [B, C, ...]: eval mode (running statistics) unless inside train_mode()."""
    shape = (1, -1) + (1,) * (x.ndim - 2)
    if _TRAIN is not None:
        dims = (0,) + tuple(range(2, x.ndim))
        n = x.numel() // x.shape[1]
        if n <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        with torch.no_grad():
            _TRAIN[pre + "running_mean"] = 0.9 * sd[pre + "running_mean"] + 0.1 * mean.detach()
            _TRAIN[pre + "running_var"] = 0.9 * sd[pre + "running_var"] + 0.1 * var.detach() * (n / (n - 1))
            _TRAIN[pre + "num_batches_tracked"] = sd[pre + "num_batches_tracked"] + 1
        return (x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + 1e-5) * sd[pre + "weight"].reshape(shape) + sd[pre + "bias"].reshape(shape)
    return ((x - sd[pre + "running_mean"].reshape(shape)) / torch.sqrt(sd[pre + "running_var"].reshape(shape) + 1e-5)
            * sd[pre + "weight"].reshape(shape) + sd[pre + "bias"].reshape(shape))


def _dwconv_bn_gelu(img: Tensor, sd, pre: str) -> Tensor:
    y = F.conv2d(img, sd[pre + "dwconv.0.weight"], sd[pre + "dwconv.0.bias"], padding=1, groups=img.shape[1])
    return F.gelu(bn_eval(y, sd, pre + "dwconv.1."))


def _channel_interaction(img: Tensor, sd, pre: str) -> Tensor:
    """[B, C, H, W] -> [B, C, 1, 1] (pre-sigmoid)"""
    y = F.conv2d(img.mean(dim=(2, 3), keepdim=True), sd[pre + "channel_interaction.1.weight"], sd[pre + "channel_interaction.1.bias"])
    y = F.gelu(bn_eval(y, sd, pre + "channel_interaction.2."))
    return F.conv2d(y, sd[pre + "channel_interaction.4.weight"], sd[pre + "channel_interaction.4.bias"])


def _spatial_interaction(img: Tensor, sd, pre: str) -> Tensor:
    """[B, C, H, W] -> [B, 1, H, W] (pre-sigmoid)"""
    y = F.gelu(bn_eval(F.conv2d(img, sd[pre + "spatial_interaction.0.weight"], sd[pre + "spatial_interaction.0.bias"]), sd, pre + "spatial_interaction.1."))
    return F.conv2d(y, sd[pre + "spatial_interaction.3.weight"], sd[pre + "spatial_interaction.3.bias"])


def adaptive_spatial_attention(x: Tensor, H: int, W: int, sd, pre: str, cfg: DATConfig, nH: int, shifted: bool) -> Tensor:
    """dat_arch.py:366-446.  q, k, v are zero-padded at the bottom / right to a multiple of the larger split (:376-384): the windows,
    the cyclic shift and the shift mask live on the padded _H x _W frame; a padded token is a zero vector that takes part in its
    window's softmax with score = bias, and the padded rows of the result are cropped (:409-410, :415-416)."""
    B, L, C = x.shape
    s0, s1 = cfg.split_size
    big = max(s0, s1)
    Hp, Wp = (H + big - 1) // big * big, (W + big - 1) // big * big
    q, k, v = F.linear(x, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias")).chunk(3, dim=-1)

    def padded(t):       # [B, L, C] -> [B, Hp * Wp, C]
        if Hp == H and Wp == W:
            return t
        return F.pad(t.reshape(B, H, W, C), (0, 0, 0, Wp - W, 0, Hp - H)).reshape(B, Hp * Wp, C)
    qp, kp, vp = padded(q), padded(k), padded(v)
    scale = cfg.qk_scale or (C // 2 // (nH // 2)) ** -0.5
    outs = []
    for br, (hs, wsz, sy, sx) in enumerate(((s0, s1, s0 // 2, s1 // 2), (s1, s0, s1 // 2, s0 // 2))):
        if not shifted:
            sy = sx = 0
        sl = slice(br * C // 2, (br + 1) * C // 2)
        idx = torch.from_numpy(rect_window_token_index(Hp, Wp, hs, wsz, sy, sx))
        nW, N = idx.shape
        hh = nH // 2

        def win(t):
            return t[:, :, sl][:, idx.reshape(-1)].reshape(B * nW, N, hh, C // 2 // hh).permute(0, 2, 1, 3)
        attn = (win(qp) * scale) @ win(kp).transpose(-2, -1) + dynamic_pos_bias(sd, f"{pre}attns.{br}.pos.", hs, wsz)[None]
        if shifted:
            attn = (attn.reshape(B, nW, hh, N, N) + torch.from_numpy(rect_shift_mask(Hp, Wp, hs, wsz, sy, sx))[None, :, None]).reshape(-1, hh, N, N)
        o = (attn.softmax(-1) @ win(vp)).transpose(1, 2).reshape(B, nW * N, C // 2)
        merged = torch.zeros(B, Hp * Wp, C // 2, dtype=x.dtype)
        merged[:, idx.reshape(-1)] = o
        outs.append(merged.reshape(B, Hp, Wp, C // 2)[:, :H, :W].reshape(B, L, C // 2))
    att = torch.cat(outs, dim=2)
    conv = _dwconv_bn_gelu(v.transpose(1, 2).reshape(B, C, H, W), sd, pre)
    cmap = _channel_interaction(conv, sd, pre).reshape(B, 1, C)
    smap = _spatial_interaction(att.transpose(1, 2).reshape(B, C, H, W), sd, pre)
    att = att * torch.sigmoid(cmap)
    conv = (torch.sigmoid(smap) * conv).flatten(2).transpose(1, 2)
    return F.linear(att + conv, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def adaptive_channel_attention(x: Tensor, H: int, W: int, sd, pre: str, cfg: DATConfig, nH: int) -> Tensor:
    B, N, C = x.shape
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias")).reshape(B, N, 3, nH, C // nH).permute(2, 0, 3, 4, 1)   # [3,B,h,d,N]
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * sd[pre + "temperature"]
    att = (attn.softmax(-1) @ v).permute(0, 3, 1, 2).reshape(B, N, C)
    conv = _dwconv_bn_gelu(v.reshape(B, C, N).reshape(B, C, H, W), sd, pre)
    cmap = _channel_interaction(att.transpose(1, 2).reshape(B, C, H, W), sd, pre)
    smap = _spatial_interaction(conv, sd, pre).permute(0, 2, 3, 1).reshape(B, N, 1)
    att = att * torch.sigmoid(smap)
    conv = (conv * torch.sigmoid(cmap)).flatten(2).transpose(1, 2)
    return F.linear(att + conv, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def sgfn(x: Tensor, H: int, W: int, sd, pre: str) -> Tensor:
    B, N, _ = x.shape
    h = F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
    x1, x2 = h.chunk(2, dim=-1)
    Ch = x2.shape[-1]
    x2 = F.layer_norm(x2, (Ch,), sd[pre + "sg.norm.weight"], sd[pre + "sg.norm.bias"], 1e-5)
    x2 = F.conv2d(x2.transpose(1, 2).reshape(B, Ch, H, W), sd[pre + "sg.conv.weight"], sd[pre + "sg.conv.bias"], padding=1, groups=Ch)
    return F.linear(x1 * x2.flatten(2).transpose(1, 2), sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def datb(x: Tensor, H: int, W: int, sd, pre: str, cfg: DATConfig, nH: int, rg: int, b: int, drop: Optional[Tensor] = None) -> Tensor:
    """drop: None or [2, B] DropPath factors (0 or 1 / keep) of the attention and the FFN branch (:562-563, drawn per sample)."""
    C = x.shape[-1]
    xn = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    if b % 2 == 0:
        a = adaptive_spatial_attention(xn, H, W, sd, pre + "attn.", cfg, nH, is_shifted(rg, b))
    else:
        a = adaptive_channel_attention(xn, H, W, sd, pre + "attn.", cfg, nH)
    x = x + (a if drop is None else a * drop[0].reshape(-1, 1, 1))
    f = sgfn(F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5), H, W, sd, pre + "ffn.")
    return x + (f if drop is None else f * drop[1].reshape(-1, 1, 1))


def forward_features(f: Tensor, sd, cfg: DATConfig, drop: Optional[Tensor] = None) -> Tensor:
    B, C, H, W = f.shape
    x = F.layer_norm(f.flatten(2).transpose(1, 2), (C,), sd["before_RG.1.weight"], sd["before_RG.1.bias"], 1e-5)
    k = 0
    for li, (depth, nH) in enumerate(zip(cfg.depth, cfg.num_heads)):
        y = x
        for bi in range(depth):
            y = datb(y, H, W, sd, f"layers.{li}.blocks.{bi}.", cfg, nH, li, bi, None if drop is None else drop[k])
            k += 1
        y = O._resi_conv(y.transpose(1, 2).reshape(B, C, H, W), sd, f"layers.{li}.conv", cfg.resi_connection)
        x = x + y.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    return x.transpose(1, 2).reshape(B, C, H, W)


def dat_forward(sd: Dict[str, Tensor], cfg: DATConfig, x: Tensor, drop: Optional[Tensor] = None) -> Tensor:
    mean = torch.tensor([0.4488, 0.4371, 0.4040], dtype=x.dtype).reshape(1, 3, 1, 1) if cfg.in_chans == 3 else torch.zeros(1, 1, 1, 1, dtype=x.dtype)
    x = (x - mean) * cfg.img_range
    s = cfg.upscale
    f = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    f = O._resi_conv(forward_features(f, sd, cfg, drop), sd, "conv_after_body", cfg.resi_connection) + f
    if cfg.upsampler == "pixelshuffle":
        f = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], padding=1), 0.01)
        if s & (s - 1) == 0:
            for i in range(int(math.log2(s))):
                f = O.pixel_shuffle(F.conv2d(f, sd[f"upsample.{2 * i}.weight"], sd[f"upsample.{2 * i}.bias"], padding=1), 2)
        elif s == 3:
            f = O.pixel_shuffle(F.conv2d(f, sd["upsample.0.weight"], sd["upsample.0.bias"], padding=1), 3)
        else:
            raise ValueError(f"scale {s} is not supported. Supported scales: 2^n and 3.")
        x = F.conv2d(f, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    elif cfg.upsampler == "pixelshuffledirect":
        x = O.pixel_shuffle(F.conv2d(f, sd["upsample.0.weight"], sd["upsample.0.bias"], padding=1), s)
    return x / cfg.img_range + mean


def loss_and_grads(sd: Dict[str, Tensor], cfg: DATConfig, x: Tensor, target: Tensor, drop: Optional[Tensor] = None):
    """One training step's numbers: L1 loss (the training script's criterion), d loss / d parameter for every floating-point
    non-buffer entry, and the BatchNorm running statistics after the step.  drop: [n_blocks, 2, B] DropPath factors or None."""
    leaf = {}
    for k, v in sd.items():
        is_param = v.is_floating_point() and not (k.endswith("running_mean") or k.endswith("running_var") or "rpe_biases" in k or "attn_mask" in k)
        leaf[k] = v.detach().clone().requires_grad_(True) if is_param else v
    record: dict = {}
    with train_mode(record):
        out = dat_forward(leaf, cfg, x, drop)
    loss = (out - target).abs().mean()
    names = [k for k, v in leaf.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    return float(loss.detach()), out.detach(), {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, grads)}, record


# ---- state_dict schema + deterministic weights -------------------------------------------------------------------------------
def state_dict_schema(cfg: DATConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) in the reference's state_dict() order; kind in {w, b, bn_w, bn_b, bn_mean, bn_var, bn_n, buf_*, ...}."""
    C = cfg.embed_dim
    hid = int(C * cfg.expansion_factor)
    s0, s1 = cfg.split_size
    res = cfg.img_size
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def lin(n, o, i, bias=True):
        out.append((n + ".weight", (o, i), "w"))
        if bias:
            out.append((n + ".bias", (o,), "b"))

    def conv(n, o, i, k=3):
        out.extend([(n + ".weight", (o, i, k, k), "w"), (n + ".bias", (o,), "b")])

    def ln(n, c):
        out.extend([(n + ".weight", (c,), "ln_w"), (n + ".bias", (c,), "ln_b")])

    def bn(n, c):
        out.extend([(n + ".weight", (c,), "ln_w"), (n + ".bias", (c,), "ln_b"), (n + ".running_mean", (c,), "bn_mean"),
                    (n + ".running_var", (c,), "bn_var"), (n + ".num_batches_tracked", (), "bn_n")])

    def interactions(p):
        conv(p + "dwconv.0", C, 1)
        bn(p + "dwconv.1", C)
        conv(p + "channel_interaction.1", C // 8, C, 1)
        bn(p + "channel_interaction.2", C // 8)
        conv(p + "channel_interaction.4", C, C // 8, 1)
        conv(p + "spatial_interaction.0", C // 16, C, 1)
        bn(p + "spatial_interaction.1", C // 16)
        conv(p + "spatial_interaction.3", 1, C // 16, 1)

    conv("conv_first", C, cfg.in_chans)
    ln("before_RG.1", C)
    for li, (depth, nH) in enumerate(zip(cfg.depth, cfg.num_heads)):
        for bi in range(depth):
            p = f"layers.{li}.blocks.{bi}."
            ln(p + "norm1", C)
            a = p + "attn."
            if bi % 2 == 0:
                if is_shifted(li, bi):
                    nW = (res // s0) * (res // s1)
                    out.append((a + "attn_mask_0", (nW, s0 * s1, s0 * s1), "buf_mask0"))
                    out.append((a + "attn_mask_1", (nW, s0 * s1, s0 * s1), "buf_mask1"))
                lin(a + "qkv", 3 * C, C, cfg.qkv_bias)
                lin(a + "proj", C, C)
                pd = (C // 2 // 4) // 4
                for br, (hs, wsz) in enumerate(((s0, s1), (s1, s0))):
                    q = f"{a}attns.{br}."
                    out.append((q + "rpe_biases", ((2 * hs - 1) * (2 * wsz - 1), 2), f"buf_rpe{br}"))
                    out.append((q + "relative_position_index", (hs * wsz, hs * wsz), f"buf_rpi{br}"))
                    lin(q + "pos.pos_proj", pd, 2)
                    for k in ("pos1", "pos2"):
                        ln(f"{q}pos.{k}.0", pd)
                        lin(f"{q}pos.{k}.2", pd, pd)
                    ln(q + "pos.pos3.0", pd)
                    lin(q + "pos.pos3.2", nH // 2, pd)
                interactions(a)
            else:
                out.append((a + "temperature", (nH, 1, 1), "temp"))
                lin(a + "qkv", 3 * C, C, cfg.qkv_bias)
                lin(a + "proj", C, C)
                interactions(a)
            lin(p + "ffn.fc1", hid, C)
            ln(p + "ffn.sg.norm", hid // 2)
            conv(p + "ffn.sg.conv", hid // 2, 1)
            lin(p + "ffn.fc2", C, hid // 2)
            ln(p + "norm2", C)
        if cfg.resi_connection == "1conv":
            conv(f"layers.{li}.conv", C, C)
        else:
            conv(f"layers.{li}.conv.0", C // 4, C)
            conv(f"layers.{li}.conv.2", C // 4, C // 4, 1)
            conv(f"layers.{li}.conv.4", C, C // 4)
    ln("norm", C)
    if cfg.resi_connection == "1conv":
        conv("conv_after_body", C, C)
    else:
        conv("conv_after_body.0", C // 4, C)
        conv("conv_after_body.2", C // 4, C // 4, 1)
        conv("conv_after_body.4", C, C // 4)
    if cfg.upsampler == "pixelshuffle":
        conv("conv_before_upsample.0", 64, C)
        s = cfg.upscale
        if s & (s - 1) == 0:
            for i in range(int(math.log2(s))):
                conv(f"upsample.{2 * i}", 256, 64)
        elif s == 3:
            conv("upsample.0", 576, 64)
        conv("conv_last", cfg.in_chans, 64)
    elif cfg.upsampler == "pixelshuffledirect":
        conv("upsample.0", cfg.upscale ** 2 * cfg.in_chans, C)
    return out


def random_state_dict(cfg: DATConfig, seed: int = 42, scale: float = 1.0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    s0, s1 = cfg.split_size
    res = cfg.img_size
    sd: Dict[str, Tensor] = {}
    for key, shape, kind in state_dict_schema(cfg):
        if kind.startswith("buf_mask"):
            hs, wsz = (s0, s1) if kind.endswith("0") else (s1, s0)
            sd[key] = torch.from_numpy(rect_shift_mask(res, res, hs, wsz, hs // 2, wsz // 2))
        elif kind.startswith("buf_rpe"):
            hs, wsz = (s0, s1) if kind.endswith("0") else (s1, s0)
            sd[key] = torch.from_numpy(rpe_offsets(hs, wsz))
        elif kind.startswith("buf_rpi"):
            hs, wsz = (s0, s1) if kind.endswith("0") else (s1, s0)
            sd[key] = torch.from_numpy(rect_rpi(hs, wsz))
        elif kind == "bn_n":
            sd[key] = torch.tensor(0, dtype=torch.int64)
        elif kind == "bn_var":
            sd[key] = torch.rand(shape, generator=g) * 0.6 + 0.7
        elif kind == "bn_mean":
            sd[key] = 0.1 * torch.randn(shape, generator=g)
        elif kind == "ln_w":
            sd[key] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "ln_b":
            sd[key] = 0.05 * torch.randn(shape, generator=g)
        elif kind == "temp":
            sd[key] = 1.0 + 0.5 * torch.rand(shape, generator=g)
        elif len(shape) == 4:
            fan = shape[1] * shape[2] * shape[3]
            sd[key] = (torch.rand(shape, generator=g) * 2 - 1) * (scale / math.sqrt(fan))
        elif kind == "b":
            sd[key] = 0.02 * torch.randn(shape, generator=g)
        elif ".pos." in key:
            sd[key] = 0.5 * torch.randn(shape, generator=g)           # the 5-wide position MLP: O(1) weights so the bias matters
        else:
            sd[key] = 0.02 * scale * torch.randn(shape, generator=g)
    return sd
