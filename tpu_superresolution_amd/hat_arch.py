"""HAT with the reference's constructor, module tree and state_dict, executed by libsrk kernels on MI355X.

Drop-in for ``modules/hat_arch.py`` of ViacheslavTimofeev/tpu_superresolution: same ``HAT(...)`` keyword arguments
(hat_arch.py:738-764), same parameter / buffer names and shapes (864 keys / 20 772 507 parameters for the HAT-SRx4
configuration: public ``.pth`` files load with ``strict=True``), same ``forward(x[B,C,H,W]) -> [B,C,H*s,W*s]``.

The reference is Python calling stock ATen operators; here the module tree only holds parameters and ``forward`` is a
host-side sequence of C-ABI calls (``include/srk.h``), one per fused kernel -- PyTorch supplies device memory and the
stream, nothing else:

    check_image_size + normalise      srk_img_prep              (:963-975)
    conv_first                        srk_stem_conv             (fp32)
    every LayerNorm                   srk_layernorm_fwd / fused into the producing GEMM's epilogue
    qkv / proj / fc1 / fc2            srk_gemm_ex (persistent LDS-DMA GEMMs), Mlp as ONE kernel (srk_mlp_fused_fwd) at width 180
    (S)W-MSA 16x16 and OCAB           srk_win256_attention_fwd: roll / partition / unfold / reverse folded into the addresses,
                                      shift mask arithmetic, bias table indexed in-kernel (closed-form rpi, negative rpi_oca wrapped)
    CAB                               two implicit-GEMM 3x3 convs (+GELU), srk_channel_gate, srk_cab_add_ln (+ norm2)
    RHAG conv, conv_after_body, head  implicit-GEMM 3x3 convs with residual / LeakyReLU / PixelShuffle / image epilogues

Scope: inference AND training.  window_size 16 (the attention kernels hold 256 queries per window),
head_dim <= 32, embed_dim <= 256, upsampler 'pixelshuffle' (the only head the reference's forward implements, :976-985),
resi_connection '1conv'.  A grad-enabled forward is ONE autograd node (``hat_train.HATFunction``): its forward keeps every
block's activations, its backward is the hand-written backward sequence of ``hat_train.hat_backward`` (256-query window
attention backward incl. the table gradient through the wrapped relative_position_index_OCA, CAB / channel-attention backward,
LayerNorm / GELU / conv / linear dgrads and weight gradients); DropPath factors are drawn per block and sample and passed
to the kernels as data.  No CPU fallback: CPU tensors raise.  Weights are packed (bf16, padded, tap-major / pixel-shuffle-
permuted) once per parameter version on the device.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import GemmArgs, SrkUnsupported, check, lib


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _holder_forward(self, *a, **k):
    raise NotImplementedError(f"{type(self).__name__} only holds parameters here; run the enclosing HAT.forward")


class ChannelAttention(nn.Module):
    def __init__(self, num_feat, squeeze_factor=16):
        super().__init__()
        self.attention = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(num_feat, num_feat // squeeze_factor, 1, padding=0),
                                       nn.ReLU(inplace=True), nn.Conv2d(num_feat // squeeze_factor, num_feat, 1, padding=0), nn.Sigmoid())
    forward = _holder_forward


class CAB(nn.Module):
    def __init__(self, num_feat, compress_ratio=3, squeeze_factor=30):
        super().__init__()
        self.cab = nn.Sequential(nn.Conv2d(num_feat, num_feat // compress_ratio, 3, 1, 1), nn.GELU(),
                                 nn.Conv2d(num_feat // compress_ratio, num_feat, 3, 1, 1), ChannelAttention(num_feat, squeeze_factor))
    forward = _holder_forward


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)
    forward = _holder_forward


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)
    forward = _holder_forward


class HAB(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, compress_ratio=3, squeeze_factor=30,
                 conv_scale=0.01, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:        # hat_arch.py:252-255
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, 'shift_size must in 0-window_size'
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=_pair(self.window_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.conv_scale = conv_scale
        self.conv_block = CAB(num_feat=dim, compress_ratio=compress_ratio, squeeze_factor=squeeze_factor)
        self.drop_path_prob = float(drop_path)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
    forward = _holder_forward


class OCAB(nn.Module):
    def __init__(self, dim, input_resolution, window_size, overlap_ratio, num_heads, qkv_bias=True, qk_scale=None, mlp_ratio=2,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.window_size, self.num_heads = dim, input_resolution, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.overlap_win_size = int(window_size * overlap_ratio) + window_size
        self.norm1 = norm_layer(dim)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.unfold = nn.Unfold(kernel_size=(self.overlap_win_size, self.overlap_win_size), stride=window_size,
                                padding=(self.overlap_win_size - window_size) // 2)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((window_size + self.overlap_win_size - 1) ** 2, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=nn.GELU)
    forward = _holder_forward


class AttenBlocks(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, compress_ratio, squeeze_factor, conv_scale, overlap_ratio,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm,
                 downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = nn.ModuleList([
            HAB(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                shift_size=0 if i % 2 == 0 else window_size // 2, compress_ratio=compress_ratio, squeeze_factor=squeeze_factor,
                conv_scale=conv_scale, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer) for i in range(depth)])
        self.overlap_attn = OCAB(dim=dim, input_resolution=input_resolution, window_size=window_size, overlap_ratio=overlap_ratio,
                                 num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, mlp_ratio=mlp_ratio, norm_layer=norm_layer)
        self.downsample = None
    forward = _holder_forward


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.patches_resolution = [self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None
    forward = _holder_forward


class PatchUnEmbed(PatchEmbed):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__(img_size, patch_size, in_chans, embed_dim, None)


class RHAG(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, compress_ratio, squeeze_factor, conv_scale, overlap_ratio,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm,
                 downsample=None, use_checkpoint=False, img_size=224, patch_size=4, resi_connection='1conv'):
        super().__init__()
        self.dim, self.input_resolution = dim, input_resolution
        self.residual_group = AttenBlocks(dim=dim, input_resolution=input_resolution, depth=depth, num_heads=num_heads,
                                          window_size=window_size, compress_ratio=compress_ratio, squeeze_factor=squeeze_factor,
                                          conv_scale=conv_scale, overlap_ratio=overlap_ratio, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                          qk_scale=qk_scale, drop=drop, attn_drop=attn_drop, drop_path=drop_path, norm_layer=norm_layer,
                                          downsample=downsample, use_checkpoint=use_checkpoint)
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1) if resi_connection == '1conv' else nn.Identity()
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=dim, norm_layer=None)
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=dim, norm_layer=None)
    forward = _holder_forward


class Upsample(nn.Sequential):
    def __init__(self, scale, num_feat):
        m = []
        if (scale & (scale - 1)) == 0:
            for _ in range(int(math.log(scale, 2))):
                m += [nn.Conv2d(num_feat, 4 * num_feat, 3, 1, 1), nn.PixelShuffle(2)]
        elif scale == 3:
            m += [nn.Conv2d(num_feat, 9 * num_feat, 3, 1, 1), nn.PixelShuffle(3)]
        else:
            raise ValueError(f'scale {scale} is not supported. ' 'Supported scales: 2^n and 3.')
        super().__init__(*m)


# ---- weight packing (one-time, on the device) ----------------------------------------------------------------------------
def _rup(v, m):
    return (v + m - 1) // m * m


# ---- packing helpers -------------------------------------------------------------------------------------------------------------------
# A model is packed after every optimizer step (training), so the count of tiny torch kernels matters: inside ``batched_pack()`` the
# helpers only RECORD their request and return a placeholder; ``resolve`` groups requests of the same kind / shape / index maps (the 42
# qkv weights of HAT, ...), runs each group as ONE stack + scatter + cast and hands out views of the stacked result.  Outside the
# context they run eagerly, one tensor at a time.
_MAPS: Dict[tuple, torch.Tensor] = {}


def _cached_map(key: tuple, make):
    t = _MAPS.get(key)
    if t is None:
        t = _MAPS[key] = make()
    return t


class _Pending:
    __slots__ = ("group", "index")

    def __init__(self, group, index):
        self.group, self.index = group, index


class _Packer:
    def __init__(self):
        self.groups: Dict[tuple, list] = {}

    def add(self, kind, src, args, maps):
        key = (kind, tuple(src.shape), src.dtype, src.device, args, tuple(id(m_) if m_ is not None else None for m_ in maps))
        g = self.groups.setdefault(key, [kind, args, maps, []])
        g[3].append(src)
        return _Pending(key, len(g[3]) - 1)

    def resolve(self, P: dict) -> None:
        done = {}
        for key, (kind, args, maps, srcs) in self.groups.items():
            done[key] = _PACK_MANY[kind](torch.stack([t.float() for t in srcs]), *args, *maps)
        for name, v in P.items():
            if isinstance(v, _Pending):
                P[name] = done[v.group][v.index]


_TLS = threading.local()        # the active packer is per thread (two threads may pack two models at once)


def _active() -> Optional[_Packer]:
    return getattr(_TLS, "packer", None)


class batched_pack:
    """with batched_pack() as pk: ... P[name] = _pack_linear(...) ...; pk.resolve(P)"""

    def __enter__(self):
        self.prev = _active()
        _TLS.packer = _Packer()
        return _TLS.packer

    def __exit__(self, *exc):
        _TLS.packer = self.prev


def _many_linear(ws, NP, KP, row_map, col_map):
    n, N, K = ws.shape
    out = torch.zeros(n, NP, KP, dtype=torch.float32, device=ws.device)
    rows = row_map if row_map is not None else torch.arange(N, device=ws.device)
    cols = col_map if col_map is not None else torch.arange(K, device=ws.device)
    out[:, rows[:, None], cols[None, :]] = ws
    return out.to(torch.bfloat16)


def _many_vec(bs, NP, row_map):
    n, N = bs.shape
    out = torch.zeros(n, NP, dtype=torch.float32, device=bs.device)
    rows = row_map if row_map is not None else torch.arange(N, device=bs.device)
    out[:, rows] = bs
    return out


def _many_conv(ws, NP, CinP, row_map):
    n, Cout, Cin = ws.shape[:3]
    out = torch.zeros(n, NP, 9, CinP, dtype=torch.float32, device=ws.device)
    rows = row_map if row_map is not None else torch.arange(Cout, device=ws.device)
    out[:, rows, :, :Cin] = ws.permute(0, 1, 3, 4, 2).reshape(n, Cout, 9, Cin)
    return out.reshape(n, NP, 9 * CinP).to(torch.bfloat16)


def _many_conv_T(ws, NP, CoutP, col_map):
    n, Cout, Cin = ws.shape[:3]
    out = torch.zeros(n, NP, 9, CoutP, dtype=torch.float32, device=ws.device)
    cols = col_map if col_map is not None else torch.arange(Cout, device=ws.device)
    out[:, :Cin, :, cols] = ws.flip(3, 4).permute(0, 2, 3, 4, 1).reshape(n, Cin, 9, Cout)
    return out.reshape(n, NP, 9 * CoutP).to(torch.bfloat16)


_PACK_MANY = {"linear": _many_linear, "vec": _many_vec, "conv": _many_conv, "convT": _many_conv_T}


def _pack_linear(w: torch.Tensor, NP: int, KP: int, row_map=None, col_map=None) -> torch.Tensor:
    """fp32 [N][K] -> bf16 [NP][KP], rows / columns scattered through the given index maps (zero elsewhere)."""
    pk = _active()
    if pk is not None:
        return pk.add("linear", w, (NP, KP), (row_map, col_map))
    return _many_linear(w.float()[None], NP, KP, row_map, col_map)[0].contiguous()


def _pack_vec(b: Optional[torch.Tensor], NP: int, row_map=None, device=None) -> torch.Tensor:
    if b is None:
        return torch.zeros(NP, dtype=torch.float32, device=device)
    pk = _active()
    if pk is not None:
        return pk.add("vec", b, (NP,), (row_map,))
    return _many_vec(b.float()[None], NP, row_map)[0]


def _pack_conv(w: torch.Tensor, NP: int, CinP: int, row_map=None) -> torch.Tensor:
    """[Cout][Cin][3][3] -> bf16 [NP][9 * CinP], K tap-major: k = (3 ky + kx) * CinP + ci."""
    pk = _active()
    if pk is not None:
        return pk.add("conv", w, (NP, CinP), (row_map,))
    return _many_conv(w.float()[None], NP, CinP, row_map)[0].contiguous()


def _pack_conv_T(w: torch.Tensor, NP: int, CoutP: int, col_map=None) -> torch.Tensor:
    """[Cout][Cin][3][3] -> bf16 [NP (input channels)][9 * CoutP]: the dgrad's weight, taps flipped, K = (tap, output channel)."""
    pk = _active()
    if pk is not None:
        return pk.add("convT", w, (NP, CoutP), (col_map,))
    return _many_conv_T(w.float()[None], NP, CoutP, col_map)[0].contiguous()


def _head_map(nH: int, dh: int, device) -> torch.Tensor:
    """channel h * dh + d -> padded channel h * 32 + d   (one tensor per (heads, head_dim, device): the batched packer groups by identity)"""
    def make():
        c = torch.arange(nH * dh, device=device)
        return (c // dh) * 32 + c % dh
    return _cached_map(("head", nH, dh, str(device)), make)


def _qkv_rows(nH: int, dh: int, device) -> torch.Tensor:
    """row of the packed qkv weight of output feature (which, h, d)"""
    return _cached_map(("qkv", nH, dh, str(device)), lambda: torch.cat([w * nH * 32 + _head_map(nH, dh, device) for w in range(3)]))


def _ps_map(C_out: int, r: int, Cs: int, device) -> torch.Tensor:
    """PixelShuffle conv: original output channel c * r^2 + i * r + j -> packed row (i * r + j) * Cs + c"""
    def make():
        n = torch.arange(C_out, device=device)
        c, ij = n // (r * r), n % (r * r)
        return ij * Cs + c
    return _cached_map(("ps", C_out, r, Cs, str(device)), make)


class HAT(nn.Module):
    r"""Hybrid Attention Transformer -- reference constructor signature, hat_arch.py:738-764."""

    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=(6, 6, 6, 6), num_heads=(6, 6, 6, 6), window_size=7,
                 compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_checkpoint=False, upscale=2, img_range=1., upsampler='', resi_connection='1conv', **kwargs):
        super().__init__()
        self.window_size, self.shift_size, self.overlap_ratio = window_size, window_size // 2, overlap_ratio
        num_feat = 64
        self.img_range, self.upscale, self.upsampler = img_range, upscale, upsampler
        self.in_chans, self.embed_dim, self.num_features, self.mlp_ratio = in_chans, embed_dim, embed_dim, mlp_ratio
        self.depths, self.heads = list(depths), list(num_heads)
        self.num_layers, self.ape, self.patch_norm = len(depths), ape, patch_norm
        self.qkv_bias, self.qk_scale, self.resi_connection, self.patch_size = qkv_bias, qk_scale, resi_connection, patch_size
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        self.compress_ratio, self.squeeze_factor, self.conv_scale = compress_ratio, squeeze_factor, conv_scale
        self.mean = torch.Tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1) if in_chans == 3 else torch.zeros(1, 1, 1, 1)
        self.register_buffer('relative_position_index_SA', self.calculate_rpi_sa())
        self.register_buffer('relative_position_index_OCA', self.calculate_rpi_oca())
        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        self.patches_resolution = self.patch_embed.patches_resolution
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim, embed_dim=embed_dim)
        if ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(RHAG(dim=embed_dim, input_resolution=tuple(self.patches_resolution), depth=depths[i], num_heads=num_heads[i],
                                    window_size=window_size, compress_ratio=compress_ratio, squeeze_factor=squeeze_factor,
                                    conv_scale=conv_scale, overlap_ratio=overlap_ratio, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                                    drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer, downsample=None,
                                    use_checkpoint=use_checkpoint, img_size=img_size, patch_size=patch_size,
                                    resi_connection=resi_connection))
        self.norm = norm_layer(self.num_features)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1) if resi_connection == '1conv' else nn.Identity()
        if upsampler == 'pixelshuffle':
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = Upsample(upscale, num_feat)
            self.conv_last = nn.Conv2d(num_feat, in_chans, 3, 1, 1)
        self.apply(self._init_weights)
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_version = -1
        self._packed_device = None

    # -- reference helper API -------------------------------------------------------------------------------------------
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def calculate_rpi_sa(self):
        """hat_arch.py:881-894 in closed form."""
        ws = self.window_size
        y, x = torch.arange(ws).repeat_interleave(ws), torch.arange(ws).repeat(ws)
        return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)

    def calculate_rpi_oca(self):
        """hat_arch.py:896-918 in closed form; negative for part of its range (the reference indexes with it as-is)."""
        ws = self.window_size
        wse = ws + int(self.overlap_ratio * ws)
        yp, xp = torch.arange(ws).repeat_interleave(ws), torch.arange(ws).repeat(ws)
        yk, xk = torch.arange(wse).repeat_interleave(wse), torch.arange(wse).repeat(wse)
        off = ws - wse + 1
        return (yk[None, :] - yp[:, None] + off) * (ws + wse - 1) + (xk[None, :] - xp[:, None] + off)

    def calculate_mask(self, x_size):
        """[nW, N, N] in {0, -100} (hat_arch.py:921-941); the attention kernel evaluates the same labels arithmetically."""
        from .network_swinir import _shift_mask_cpu
        return _shift_mask_cpu(x_size[0], x_size[1], self.window_size, self.shift_size)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def check_image_size(self, x):
        _, _, h, w = x.size()
        ph = (self.window_size - h % self.window_size) % self.window_size
        pw = (self.window_size - w % self.window_size) % self.window_size
        return torch.nn.functional.pad(x, (0, pw, 0, ph), 'reflect')

    # -- packing --------------------------------------------------------------------------------------------------------
    def _unsupported_reason(self) -> Optional[str]:
        C = self.embed_dim
        if self.upsampler != 'pixelshuffle':
            return f"upsampler={self.upsampler!r} (the reference's forward only implements 'pixelshuffle')"
        if self.window_size != 16:
            return f"window_size={self.window_size} (the attention kernels hold 16 x 16 = 256 queries per window)"
        if min(self.patches_resolution) <= self.window_size:
            return "img_size <= window_size"
        if self.resi_connection != '1conv' or self.ape or not self.patch_norm or not self.qkv_bias or self.patch_size != 1:
            return "resi_connection / ape / patch_norm / qkv_bias / patch_size other than the defaults"
        if self.drop_rate or self.attn_drop_rate:
            return "dropout > 0"
        if C > 256 or any(C % h or C // h > 32 for h in self.heads):
            return "embed_dim > 256 or head_dim > 32"
        if int(self.overlap_ratio * self.window_size) != 8:
            return f"overlap_ratio={self.overlap_ratio} (the overlapping cross-attention kernel is built for 24 x 24 key windows)"
        if C // self.compress_ratio > 64 or not 1 <= C // self.squeeze_factor <= 64:
            return "compress_ratio / squeeze_factor out of the kernels' range"
        if self.in_chans not in (1, 3):
            return f"in_chans={self.in_chans}"
        return None

    def _pack(self, device) -> Dict[str, torch.Tensor]:
        """bf16 / padded / permuted copies of the parameters in the kernels' layouts, rebuilt when a parameter changed."""
        ver = sum(p._version for p in self.parameters())
        if self._packed is not None and self._packed_version == ver and self._packed_device == device:
            return self._packed
        C, CP = self.embed_dim, _rup(self.embed_dim, 64)
        hid = int(C * self.mlp_ratio)
        HP = _rup(hid, 64)
        P: Dict[str, torch.Tensor] = {}
        with torch.no_grad(), batched_pack() as pk:
            for li, layer in enumerate(self.layers):
                nH = self.heads[li]
                dh, CA = C // nH, nH * 32
                hm = _head_map(nH, dh, device)
                qkv_rows = _qkv_rows(nH, dh, device)

                def attn_pack(pre, qkv, proj):
                    P[pre + "Wqkv"] = _pack_linear(qkv.weight, 3 * CA, CP, row_map=qkv_rows)
                    P[pre + "bqkv"] = _pack_vec(qkv.bias, 3 * CA, row_map=qkv_rows)
                    P[pre + "Wproj"] = _pack_linear(proj.weight, CP, CA, col_map=hm)
                    P[pre + "bproj"] = _pack_vec(proj.bias, CP)

                def mlp_pack(pre, mlp):
                    P[pre + "W1"] = _pack_linear(mlp.fc1.weight, HP, CP)
                    P[pre + "b1"] = _pack_vec(mlp.fc1.bias, HP)
                    P[pre + "W2"] = _pack_linear(mlp.fc2.weight, CP, HP)
                    P[pre + "b2"] = _pack_vec(mlp.fc2.bias, CP)

                for bi, blk in enumerate(layer.residual_group.blocks):
                    pre = f"{li}.{bi}."
                    attn_pack(pre, blk.attn.qkv, blk.attn.proj)
                    mlp_pack(pre, blk.mlp)
                    cab = blk.conv_block.cab
                    P[pre + "Wc0"] = _pack_conv(cab[0].weight, 64, CP)
                    P[pre + "bc0"] = _pack_vec(cab[0].bias, 64)
                    P[pre + "Wc2"] = _pack_conv(cab[2].weight, CP, 64)
                    P[pre + "bc2"] = _pack_vec(cab[2].bias, CP)
                    att = cab[3].attention
                    P[pre + "ca_w1"] = att[1].weight.float().reshape(att[1].weight.shape[0], C).contiguous()
                    P[pre + "ca_b1"] = att[1].bias.float().contiguous()
                    P[pre + "ca_w2"] = att[3].weight.float().reshape(C, att[3].weight.shape[1]).contiguous()
                    P[pre + "ca_b2"] = att[3].bias.float().contiguous()
                oc = layer.residual_group.overlap_attn
                pre = f"{li}.oca."
                attn_pack(pre, oc.qkv, oc.proj)
                mlp_pack(pre, oc.mlp)
                P[f"{li}.Wconv"] = _pack_conv(layer.conv.weight, CP, CP)
                P[f"{li}.bconv"] = _pack_vec(layer.conv.bias, CP)
            P["Wcab"] = _pack_conv(self.conv_after_body.weight, CP, CP)
            P["bcab"] = _pack_vec(self.conv_after_body.bias, CP)
            P["Wbefore"] = _pack_conv(self.conv_before_upsample[0].weight, 64, CP)
            P["bbefore"] = _pack_vec(self.conv_before_upsample[0].bias, 64)
            k = 0
            for m in self.upsample:
                if isinstance(m, nn.Conv2d):
                    r = int(round(math.sqrt(m.weight.shape[0] // 64)))
                    pm = _ps_map(m.weight.shape[0], r, 64, device)
                    P[f"Wup{k}"] = _pack_conv(m.weight, m.weight.shape[0], 64, row_map=pm)
                    P[f"bup{k}"] = _pack_vec(m.bias, m.weight.shape[0], row_map=pm)
                    P[f"rup{k}"] = torch.tensor(r)
                    k += 1
            P["Wlast"] = _pack_conv(self.conv_last.weight, 16, 64)
            P["blast"] = _pack_vec(self.conv_last.bias, 16)
            pk.resolve(P)
        self._packed, self._packed_version, self._packed_device = P, ver, device
        return P

    # -- forward -----------------------------------------------------------------------------------------------------------
    def forward_features(self, x):
        raise NotImplementedError("forward_features is part of HAT.forward on the HIP path")

    def draw_drop_path(self, B: int, device) -> Optional[torch.Tensor]:
        """DropPath factors [n_blocks][2][B] (0 or 1 / keep; hat_arch.py:258 draws them per sample) or None when every rate is 0"""
        probs = [blk.drop_path_prob for layer in self.layers for blk in layer.residual_group.blocks]
        if not any(pr > 0 for pr in probs):
            return None
        keep = getattr(self, "_keep_cache", None)          # built once per device: a host -> device upload cannot be graph-captured
        if keep is None or keep.device != torch.device(device):
            keep = self._keep_cache = 1.0 - torch.tensor(probs, dtype=torch.float32, device=device).view(-1, 1, 1)
        return (torch.rand(len(probs), 2, B, device=device) < keep).float() / keep

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("this HAT runs on MI355X through libsrk only; move the model and input to the GPU (no CPU fallback)")
        why = self._unsupported_reason()
        if why:
            raise SrkUnsupported(f"the MI355X HIP path does not cover {why}; no fallback path exists in this package")
        p0 = next(self.parameters())
        if p0.device != x.device:
            raise RuntimeError(f"input is on {x.device} but the model is on {p0.device}")
        _lib.claim_device(x.device.index if x.device.index is not None else torch.cuda.current_device())
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training (or a grad-enabled eval forward): one autograd node whose backward is the C-ABI backward sequence of
            # hat_train.py; DropPath factors are drawn here (train mode, hat_arch.py:258) and passed in as data
            from .hat_train import HATFunction
            drop = None
            if self.training:
                drop = getattr(self, "_drop_override", None)      # training.GraphedTrainStep draws the factors outside its graph
                if drop is None:
                    drop = self.draw_drop_path(x.shape[0], x.device)
            return HATFunction.apply(self, x, drop, *[p for _, p in self.named_parameters()])
        with torch.no_grad(), torch.cuda.device(x.device):
            return _hat_forward(self, x.contiguous().float(), self._pack(x.device))


# ---- the launch sequence ----------------------------------------------------------------------------------------------------
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _gemm(st, loader, ep, A, W, M, N, K, *, lda=0, conv=None, bias=None, outf=None, outb=None, outb2=None, res=None, ldo=0, scale=0.0,
          r=0, Cs=0, img=None, xn=None, aux=None, rowscale=None, rows_per_sample=0, ln=None):
    a = GemmArgs()
    a.loader, a.epilogue = loader, ep
    a.A, a.lda, a.W, a.M, a.N, a.K = _ptr(A), lda, _ptr(W), M, N, K
    if conv is not None:
        a.B, a.H, a.Wd, a.CinP = conv
    a.r, a.Cs = r, Cs
    a.bias, a.outf, a.outb, a.outb2, a.res = _ptr(bias), _ptr(outf), _ptr(outb), _ptr(outb2), _ptr(res)
    a.aux, a.rowscale, a.rows_per_sample = _ptr(aux), _ptr(rowscale), (rows_per_sample if rowscale is not None else 0)
    a.ldo, a.scale = ldo or N, scale
    if img is not None:
        a.inv_range, a.Cimg, a.Hc, a.Wc = img["inv_range"], img["Cimg"], img["Hc"], img["Wc"]
        for i in range(4):
            a.mean[i] = img["mean"][i]
    if xn is not None:
        a.xn_out, a.xn_mean, a.xn_rstd, a.xn_gamma, a.xn_beta, a.xn_C = (_ptr(xn["out"]), _ptr(xn["mean"]), _ptr(xn["rstd"]),
                                                                       _ptr(xn["gamma"]), _ptr(xn["beta"]), xn["C"])
    if ln is not None:      # EP_LNBWD: LayerNorm backward fused into the dgrad's epilogue
        a.ln_x, a.ln_mean, a.ln_rstd, a.ln_gamma = _ptr(ln["x"]), _ptr(ln["mean"]), _ptr(ln["rstd"]), _ptr(ln["gamma"])
        a.ln_dgamma, a.ln_dbeta, a.ln_C = _ptr(ln["dgamma"]), _ptr(ln["dbeta"]), ln["C"]
    check(lib().srk_gemm_ex(C.byref(a), st))


def _hat_forward(m: HAT, x: torch.Tensor, P: Dict[str, torch.Tensor]) -> torch.Tensor:
    dev = x.device
    st = torch.cuda.current_stream(dev).cuda_stream
    B, Cin, H0, W0 = x.shape
    ws, s = m.window_size, m.upscale
    H, W = _rup(H0, ws), _rup(W0, ws)
    if (H - H0 >= H0) or (W - W0 >= W0):
        raise RuntimeError(f"reflect padding {H0}x{W0} -> {H}x{W} needs pad < size (as torch 'reflect')")
    T, HW = B * H * W, H * W
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    hid = int(C_ * m.mlp_ratio)
    HP = _rup(hid, 64)
    f32 = dict(dtype=torch.float32, device=dev)
    b16 = dict(dtype=torch.bfloat16, device=dev)
    L = lib()

    mean3 = (C.c_float * 3)(*(m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]))
    img4 = torch.empty(T, 4, **f32)
    check(L.srk_img_prep(x.data_ptr(), img4.data_ptr(), B, Cin, H0, W0, H, W, float(m.img_range), C.byref(mean3), st))
    f0 = torch.empty(T, CP, **f32)
    check(L.srk_stem_conv(img4.data_ptr(), m.conv_first.weight.data_ptr(), m.conv_first.bias.data_ptr(), f0.data_ptr(), B, H, W, Cin, C_, CP, st))
    _, cur, _, _ = ops.layernorm_fwd(f0, m.patch_embed.norm.weight, m.patch_embed.norm.bias, C_, out_bf16=False, out_f32=True)

    # scratch shared by all blocks
    qkv = torch.empty(T, 3 * max(h * 32 for h in m.heads), **b16)
    ao = torch.empty(T, max(h * 32 for h in m.heads), **b16)
    c1 = torch.empty(T, 64, **b16)
    c2 = torch.empty(T, CP, **b16)
    gate = torch.empty(B, CP, **f32)
    gate_ws = torch.empty(max(1, int(L.srk_channel_gate_workspace(B, HW, CP))), dtype=torch.uint8, device=dev)
    xn2 = torch.empty(T, CP, **b16)
    hh = torch.empty(T, HP, **b16)
    xb = torch.empty(T, CP, **b16)
    stat_a, stat_b = torch.empty(T, **f32), torch.empty(T, **f32)
    fused_mlp_ok = (CP == 192 and HP == 384 and T % 64 == 0 and T >= 64 * torch.cuda.get_device_properties(dev).multi_processor_count)

    ln_fusable = CP in (64, 128, 192)          # the LayerNorm that consumes a freshly written row rides in the producer's epilogue
    xn_a, xn_b = torch.empty(T, CP, **b16), torch.empty(T, CP, **b16)

    def next_norm(norm, dst):
        return dict(out=dst, mean=stat_a, rstd=stat_b, gamma=norm.weight, beta=norm.bias, C=C_) if (ln_fusable and norm is not None) else None

    def mlp(pre, xn_in, x_res, out, out_b=None, nn_=None):
        """out = x_res + fc2(gelu(fc1(xn_in)))   (Mlp.forward :86-92 + the residual add) [+ the next LayerNorm of the new rows]"""
        if fused_mlp_ok:
            args = (None, None, None, None, None, 0) if nn_ is None else (nn_["out"].data_ptr(), nn_["mean"].data_ptr(), nn_["rstd"].data_ptr(),
                                                                       nn_["gamma"].data_ptr(), nn_["beta"].data_ptr(), nn_["C"])
            check(L.srk_mlp_fused_fwd(xn_in.data_ptr(), P[pre + "W1"].data_ptr(), P[pre + "b1"].data_ptr(), P[pre + "W2"].data_ptr(),
                                      P[pre + "b2"].data_ptr(), x_res.data_ptr(), out.data_ptr(), _ptr(out_b), *args, T, st))
        else:
            _gemm(st, _lib.LD_ROWS, _lib.EP_GELU, xn_in, P[pre + "W1"], T, HP, CP, lda=CP, bias=P[pre + "b1"], outb2=hh)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, hh, P[pre + "W2"], T, CP, HP, lda=HP, bias=P[pre + "b2"], res=x_res, outf=out, outb=out_b, xn=nn_)

    xn1 = None            # norm1 of the upcoming block when the previous kernel already produced it in its epilogue
    for li, layer in enumerate(m.layers):
        nH = m.heads[li]
        CA = nH * 32
        scale = float(m.qk_scale or (C_ // nH) ** -0.5)
        layer_in = cur
        blocks = list(layer.residual_group.blocks)
        oc = layer.residual_group.overlap_attn
        for bi, blk in enumerate(blocks):
            pre = f"{li}.{bi}."
            if xn1 is None:
                xn1, _, _, _ = ops.layernorm_fwd(cur, blk.norm1.weight, blk.norm1.bias, C_)                      # norm1 :290
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
            sh = blk.shift_size
            tab = blk.attn.relative_position_bias_table        # the kernel indexes the table itself (rpi in closed form)
            check(L.srk_win256_attention_fwd(qkv.data_ptr(), 3 * CA, CA, tab.data_ptr(), tab.shape[0], ao.data_ptr(), CA, B, H, W, ws, ws,
                                             sh, sh, nH, scale, 0, st))                                           # :298-319
            x1 = torch.empty(T, CP, **f32)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, ao, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1)
            # conv branch on the un-shifted normed features (:294-295): conv3x3 + GELU, conv3x3, channel-attention gate
            _gemm(st, _lib.LD_CONV3, _lib.EP_GELU, xn1, P[pre + "Wc0"], T, 64, 9 * CP, conv=(B, H, W, CP), bias=P[pre + "bc0"], outb2=c1)
            _gemm(st, _lib.LD_CONV3, _lib.EP_BF16, c1, P[pre + "Wc2"], T, CP, 9 * 64, conv=(B, H, W, 64), bias=P[pre + "bc2"], outb=c2)
            S = P[pre + "ca_w1"].shape[0]
            check(L.srk_channel_gate(c2.data_ptr(), gate_ws.data_ptr(), P[pre + "ca_w1"].data_ptr(), P[pre + "ca_b1"].data_ptr(),
                                     P[pre + "ca_w2"].data_ptr(), P[pre + "ca_b2"].data_ptr(), float(blk.conv_scale), gate.data_ptr(), B, HW,
                                     C_, CP, S, st))
            check(L.srk_cab_add_ln(x1.data_ptr(), c2.data_ptr(), gate.data_ptr(), blk.norm2.weight.data_ptr(), blk.norm2.bias.data_ptr(),
                                   xn2.data_ptr(), T, HW, C_, CP, st))                                            # :322-323
            nxt = torch.empty(T, CP, **f32)
            following = blocks[bi + 1].norm1 if bi + 1 < len(blocks) else oc.norm1
            dst = xn_a if xn1 is not xn_a else xn_b                       # this block's xn1 is still read by nobody, but keep them apart
            nn_ = next_norm(following, dst)
            mlp(pre, xn2, x1, nxt, nn_=nn_)
            cur = nxt
            xn1 = dst if nn_ is not None else None
        pre = f"{li}.oca."                                                                                        # OCAB :389-439
        if xn1 is None:
            xn1, _, _, _ = ops.layernorm_fwd(cur, oc.norm1.weight, oc.norm1.bias, C_)
        _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
        tab = oc.relative_position_bias_table
        check(L.srk_win256_attention_fwd(qkv.data_ptr(), 3 * CA, CA, tab.data_ptr(), tab.shape[0], ao.data_ptr(), CA, B, H, W, ws, ws, 0, 0, nH,
                                         scale, oc.overlap_win_size - ws, st))
        x1 = torch.empty(T, CP, **f32)
        _gemm(st, _lib.LD_ROWS, _lib.EP_RES, ao, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1,
              xn=dict(out=xn2, mean=stat_a, rstd=stat_b, gamma=oc.norm2.weight, beta=oc.norm2.bias, C=C_))       # proj + shortcut, norm2
        x2 = torch.empty(T, CP, **f32)
        mlp(pre, xn2, x1, x2, out_b=xb)
        nxt = torch.empty(T, CP, **f32)                                                                          # RHAG :619: conv + residual
        following = m.layers[li + 1].residual_group.blocks[0].norm1 if li + 1 < len(m.layers) else m.norm      # next norm1 / final norm :958
        nn_ = next_norm(following, xn_a)
        _gemm(st, _lib.LD_CONV3, _lib.EP_RES, xb, P[f"{li}.Wconv"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P[f"{li}.bconv"], res=layer_in, outf=nxt,
              xn=nn_)
        cur = nxt
        xn1 = xn_a if nn_ is not None else None

    xnf = xn1 if xn1 is not None else ops.layernorm_fwd(cur, m.norm.weight, m.norm.bias, C_)[0]                  # norm :958
    fb = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_RES_BF16, xnf, P["Wcab"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P["bcab"], res=f0, outb=fb)
    t1 = torch.empty(T, 64, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_LRELU, fb, P["Wbefore"], T, 64, 9 * CP, conv=(B, H, W, CP), bias=P["bbefore"], outb=t1, scale=0.01)
    src, h, w = t1, H, W
    k = 0
    while f"Wup{k}" in P:
        r = int(P[f"rup{k}"])
        N = P[f"Wup{k}"].shape[0]
        up = torch.empty(B * h * r * w * r, 64, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_PS, src, P[f"Wup{k}"], B * h * w, N, 9 * 64, conv=(B, h, w, 64), bias=P[f"bup{k}"], outb=up, r=r, Cs=64, ldo=N)
        src, h, w, k = up, h * r, w * r, k + 1
    y = torch.empty(B, Cin, H0 * s, W0 * s, **f32)
    mean4 = (m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]) + [0.0]
    _gemm(st, _lib.LD_CONV3, _lib.EP_IMG, src, P["Wlast"], B * h * w, 16, 9 * 64, conv=(B, h, w, 64), bias=P["blast"], outf=y,
          img=dict(inv_range=1.0 / float(m.img_range), Cimg=Cin, Hc=H0 * s, Wc=W0 * s, mean=mean4))
    return y
