"""DAT with the reference's constructor, module tree and state_dict, executed by libsrk kernels on MI355X.

Drop-in for ``modules/dat_arch.py`` of ViacheslavTimofeev/tpu_superresolution: same ``DAT(...)`` keyword arguments
(dat_arch.py:721-741), same parameter / buffer names, shapes and order (2 116 keys / 14 802 051 parameters for the official
DAT x4 configuration), same ``forward(x[B,C,H,W]) -> [B,C,H*s,W*s]``.  As for HAT the module tree only holds parameters and
``forward`` is a host-side sequence of C-ABI calls (``include/srk.h``):

    LayerNorms, qkv / proj / fc1 / fc2, 3x3 convs, head      the SwinIR / HAT kernels (srk_layernorm_fwd, srk_gemm_ex, ...)
    Spatial_Attention (8x32 | 32x8 windows, :133-244)         srk_win256_attention_fwd once per branch on its half of the heads:
                                                              img2windows / roll / windows2img live in the kernel's addresses, the
                                                              shift mask is arithmetic, the DynamicPosBias MLP (945 x 5, :93-130) is
                                                              evaluated at pack time into a dense [heads/2][256][256] bias
    Adaptive_Channel_Attention core (:481-505)                srk_channel_attention_fwd
    DW-conv + BatchNorm + GELU branches (:310-314, :463-467)  srk_dwconv3x3 (BatchNorm's inference affine folded into scale / shift)
    channel / spatial interaction + gating (:315-327, :430-436) srk_channel_gate_act, srk_spatial_gate, srk_dual_gate_combine
    SGFN (:57-90)                                             fc1 + GELU GEMM, srk_rowln_bf16, srk_dwconv3x3 with the gating multiply, fc2 + residual

Scope: inference (eval mode, BatchNorm's running statistics folded at pack time) AND training (train mode:
batch statistics + running-statistic updates, DropPath as data, hand-written backward -- ``dat_train.py``; a grad-enabled forward
in eval mode is inference and builds no graph).  split_size with 256 or 128 tokens per window (8x32, 16x16, 8x16, ...); any H, W
(q / k / v zero-padded to a multiple of the larger split as the reference does); head_dim <= 32, embed_dim <= 256,
resi_connection '1conv', both upsamplers.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import SrkUnsupported, check, lib
from .hat_arch import (Upsample, _cached_map, _gemm, _head_map, _holder_forward, _pack_conv, _pack_linear, _pack_vec, _ps_map, _ptr, _qkv_rows, _rup,
                       batched_pack)


class UpsampleOneStep(nn.Sequential):
    def __init__(self, scale, num_feat, num_out_ch, input_resolution=None):
        self.num_feat, self.input_resolution = num_feat, input_resolution
        super().__init__(nn.Conv2d(num_feat, (scale ** 2) * num_out_ch, 3, 1, 1), nn.PixelShuffle(scale))


class SpatialGate(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.conv = nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim)
    forward = _holder_forward


class SGFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.sg = SpatialGate(hidden_features // 2)
        self.fc2 = nn.Linear(hidden_features // 2, out_features or in_features)
        self.drop = nn.Dropout(drop)
    forward = _holder_forward


class DynamicPosBias(nn.Module):
    def __init__(self, dim, num_heads, residual):
        super().__init__()
        self.residual, self.num_heads, self.pos_dim = residual, num_heads, dim // 4
        self.pos_proj = nn.Linear(2, self.pos_dim)
        self.pos1 = nn.Sequential(nn.LayerNorm(self.pos_dim), nn.ReLU(inplace=True), nn.Linear(self.pos_dim, self.pos_dim))
        self.pos2 = nn.Sequential(nn.LayerNorm(self.pos_dim), nn.ReLU(inplace=True), nn.Linear(self.pos_dim, self.pos_dim))
        self.pos3 = nn.Sequential(nn.LayerNorm(self.pos_dim), nn.ReLU(inplace=True), nn.Linear(self.pos_dim, self.num_heads))

    def forward(self, biases):          # tiny (945 x 5): evaluated with torch at pack time
        if self.residual:
            pos = self.pos_proj(biases)
            pos = pos + self.pos1(pos)
            pos = pos + self.pos2(pos)
            return self.pos3(pos)
        return self.pos3(self.pos2(self.pos1(self.pos_proj(biases))))


class Spatial_Attention(nn.Module):
    def __init__(self, dim, idx, split_size=[8, 8], dim_out=None, num_heads=6, attn_drop=0., proj_drop=0., qk_scale=None, position_bias=True):
        super().__init__()
        self.dim, self.dim_out, self.split_size, self.num_heads, self.idx = dim, dim_out or dim, split_size, num_heads, idx
        self.position_bias = position_bias
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        if idx not in (0, 1):
            raise ValueError(f"ERROR MODE {idx}")
        self.H_sp, self.W_sp = (split_size[0], split_size[1]) if idx == 0 else (split_size[1], split_size[0])
        if position_bias:
            self.pos = DynamicPosBias(self.dim // 4, self.num_heads, residual=False)
            hs, ws = self.H_sp, self.W_sp
            dy, dx = torch.arange(1 - hs, hs), torch.arange(1 - ws, ws)
            self.register_buffer('rpe_biases', torch.stack([dy.repeat_interleave(2 * ws - 1), dx.repeat(2 * hs - 1)], 1).float())
            py, px = torch.arange(hs).repeat_interleave(ws), torch.arange(ws).repeat(hs)
            self.register_buffer('relative_position_index', (py[:, None] - py[None, :] + hs - 1) * (2 * ws - 1) + (px[:, None] - px[None, :] + ws - 1))
        self.attn_drop = nn.Dropout(attn_drop)
    forward = _holder_forward


def _interaction_modules(mod: nn.Module, dim: int) -> None:
    mod.dwconv = nn.Sequential(nn.Conv2d(dim, dim, kernel_size=3, stride=1, padding=1, groups=dim), nn.BatchNorm2d(dim), nn.GELU())
    mod.channel_interaction = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(dim, dim // 8, kernel_size=1), nn.BatchNorm2d(dim // 8), nn.GELU(),
                                            nn.Conv2d(dim // 8, dim, kernel_size=1))
    mod.spatial_interaction = nn.Sequential(nn.Conv2d(dim, dim // 16, kernel_size=1), nn.BatchNorm2d(dim // 16), nn.GELU(),
                                            nn.Conv2d(dim // 16, 1, kernel_size=1))


def _rect_mask(H, W, hs, ws, sy, sx):
    def lab(n, w, s):
        v = torch.arange(n)
        return torch.where(v < n - w, 0, torch.where(v < n - s, 1, 2))
    label = (lab(H, hs, sy)[:, None] * 3 + lab(W, ws, sx)[None, :]).view(H // hs, hs, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, hs * ws)
    return torch.where(label[:, None, :] != label[:, :, None], torch.tensor(-100.0), torch.tensor(0.0))


def _fc1_rows(hid: int, half: int, HPh: int, device) -> torch.Tensor:
    """row of the packed fc1 weight of hidden feature r: the two halves of SGFN's hidden vector are each padded to HPh columns"""
    def make():
        rows = torch.arange(hid, device=device)
        return torch.where(rows < half, rows, rows - half + HPh)
    return _cached_map(("fc1", hid, half, HPh, str(device)), make)


def is_shifted(rg_idx: int, b_idx: int) -> bool:
    return (rg_idx % 2 == 0 and b_idx > 0 and (b_idx - 2) % 4 == 0) or (rg_idx % 2 != 0 and b_idx % 4 == 0)      # dat_arch.py:297


class Adaptive_Spatial_Attention(nn.Module):
    def __init__(self, dim, num_heads, reso=64, split_size=[8, 8], shift_size=[1, 2], qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 rg_idx=0, b_idx=0):
        super().__init__()
        self.dim, self.num_heads, self.split_size, self.shift_size = dim, num_heads, split_size, shift_size
        self.b_idx, self.rg_idx, self.patches_resolution = b_idx, rg_idx, reso
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        assert 0 <= self.shift_size[0] < self.split_size[0], "shift_size must in 0-split_size0"
        assert 0 <= self.shift_size[1] < self.split_size[1], "shift_size must in 0-split_size1"
        self.branch_num = 2
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(drop)
        self.attns = nn.ModuleList([Spatial_Attention(dim // 2, idx=i, split_size=split_size, num_heads=num_heads // 2, dim_out=dim // 2,
                                                      qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop, position_bias=True)
                                    for i in range(self.branch_num)])
        self.shifted = is_shifted(rg_idx, b_idx)
        if self.shifted:
            m0, m1 = self.calculate_mask(reso, reso)
            self.register_buffer("attn_mask_0", m0)
            self.register_buffer("attn_mask_1", m1)
        else:
            self.register_buffer("attn_mask_0", None)
            self.register_buffer("attn_mask_1", None)
        _interaction_modules(self, dim)

    def calculate_mask(self, H, W):
        """dat_arch.py:334-380; kept for state_dict compatibility (the attention kernel evaluates the labels arithmetically)."""
        s0, s1 = self.split_size
        return _rect_mask(H, W, s0, s1, self.shift_size[0], self.shift_size[1]), _rect_mask(H, W, s1, s0, self.shift_size[1], self.shift_size[0])
    forward = _holder_forward


class Adaptive_Channel_Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        _interaction_modules(self, dim)
    forward = _holder_forward


class DATB(nn.Module):
    def __init__(self, dim, num_heads, reso=64, split_size=[2, 4], shift_size=[1, 2], expansion_factor=4., qkv_bias=False, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, rg_idx=0, b_idx=0):
        super().__init__()
        self.norm1 = norm_layer(dim)
        if b_idx % 2 == 0:
            self.attn = Adaptive_Spatial_Attention(dim, num_heads=num_heads, reso=reso, split_size=split_size, shift_size=shift_size,
                                                   qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop, rg_idx=rg_idx, b_idx=b_idx)
        else:
            self.attn = Adaptive_Channel_Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()        # the reference's DropPath holds no state; the factors are drawn in DAT.forward and passed as data
        self.drop_path_prob = float(drop_path)
        self.ffn = SGFN(in_features=dim, hidden_features=int(dim * expansion_factor), out_features=dim, act_layer=act_layer)
        self.norm2 = norm_layer(dim)
    forward = _holder_forward


class ResidualGroup(nn.Module):
    def __init__(self, dim, reso, num_heads, split_size=[2, 4], expansion_factor=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_paths=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm, depth=2, use_chk=False, resi_connection='1conv', rg_idx=0):
        super().__init__()
        self.use_chk, self.reso = use_chk, reso
        self.blocks = nn.ModuleList([DATB(dim=dim, num_heads=num_heads, reso=reso, split_size=split_size,
                                          shift_size=[split_size[0] // 2, split_size[1] // 2], expansion_factor=expansion_factor,
                                          qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop, drop_path=drop_paths[i],
                                          act_layer=act_layer, norm_layer=norm_layer, rg_idx=rg_idx, b_idx=i) for i in range(depth)])
        if resi_connection == '1conv':
            self.conv = nn.Conv2d(dim, dim, 3, 1, 1)
        elif resi_connection == '3conv':
            self.conv = nn.Sequential(nn.Conv2d(dim, dim // 4, 3, 1, 1), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                                      nn.Conv2d(dim // 4, dim // 4, 1, 1, 0), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                                      nn.Conv2d(dim // 4, dim, 3, 1, 1))
    forward = _holder_forward


class DAT(nn.Module):
    """Dual Aggregation Transformer -- reference constructor signature, dat_arch.py:721-741."""

    def __init__(self, img_size=64, in_chans=3, embed_dim=180, split_size=[2, 4], depth=[2, 2, 2, 2], num_heads=[2, 2, 2, 2],
                 expansion_factor=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, use_chk=False, upscale=2, img_range=1., resi_connection='1conv', upsampler='pixelshuffle', **kwargs):
        super().__init__()
        num_feat = 64
        self.img_range, self.upscale, self.upsampler = img_range, upscale, upsampler
        self.in_chans, self.img_size, self.split_size = in_chans, img_size, list(split_size)
        self.mean = torch.Tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1) if in_chans == 3 else torch.zeros(1, 1, 1, 1)
        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.num_layers, self.use_chk = len(depth), use_chk
        self.num_features = self.embed_dim = embed_dim
        self.heads, self.depth, self.expansion_factor = list(num_heads), list(depth), expansion_factor
        self.qkv_bias, self.qk_scale, self.resi_connection = qkv_bias, qk_scale, resi_connection
        self.drop_rate, self.attn_drop_rate = drop_rate, attn_drop_rate
        self.before_RG = nn.Sequential(nn.Identity(), nn.LayerNorm(embed_dim))        # [0] is einops' Rearrange in the reference (no parameters)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depth))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(ResidualGroup(dim=embed_dim, num_heads=num_heads[i], reso=img_size, split_size=split_size,
                                             expansion_factor=expansion_factor, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                                             attn_drop=attn_drop_rate, drop_paths=dpr[sum(depth[:i]):sum(depth[:i + 1])], act_layer=act_layer,
                                             norm_layer=norm_layer, depth=depth[i], use_chk=use_chk, resi_connection=resi_connection, rg_idx=i))
        self.norm = norm_layer(embed_dim)
        if resi_connection == '1conv':
            self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        elif resi_connection == '3conv':
            self.conv_after_body = nn.Sequential(nn.Conv2d(embed_dim, embed_dim // 4, 3, 1, 1), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                                                 nn.Conv2d(embed_dim // 4, embed_dim // 4, 1, 1, 0), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                                                 nn.Conv2d(embed_dim // 4, embed_dim, 3, 1, 1))
        if upsampler == 'pixelshuffle':
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = Upsample(upscale, num_feat)
            self.conv_last = nn.Conv2d(num_feat, in_chans, 3, 1, 1)
        elif upsampler == 'pixelshuffledirect':
            self.upsample = UpsampleOneStep(upscale, embed_dim, in_chans, (img_size, img_size))
        self.apply(self._init_weights)
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_version = -1
        self._packed_device = None

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d, nn.GroupNorm, nn.InstanceNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # -- packing ------------------------------------------------------------------------------------------------------------
    def _unsupported_reason(self) -> Optional[str]:
        C_ = self.embed_dim
        s0, s1 = self.split_size
        if s0 * s1 not in (128, 256) or s0 % 4 or s1 % 4:
            return f"split_size={self.split_size} (the attention kernel holds 256 or 128 tokens per window, sides multiples of 4)"
        if self.resi_connection != '1conv':
            return f"resi_connection={self.resi_connection!r}"
        if self.upsampler not in ('pixelshuffle', 'pixelshuffledirect'):
            return f"upsampler={self.upsampler!r}"
        if self.upsampler == 'pixelshuffledirect' and self.upscale ** 2 * self.in_chans > 16:
            return "pixelshuffledirect with upscale^2 * in_chans > 16"
        if self.drop_rate or self.attn_drop_rate or self.in_chans not in (1, 3):
            return "dropout > 0 or in_chans not in (1, 3)"
        if C_ > 256 or any(h % 2 or C_ % h or C_ // h > 32 or h > 8 for h in self.heads):
            return "embed_dim > 256, odd / > 8 heads or head_dim > 32"
        if C_ // 16 < 1 or C_ // 16 > 16 or C_ // 8 > 64 or int(C_ * self.expansion_factor) // 2 > 512:
            return "interaction / SGFN widths out of the kernels' range"
        return None

    def _pack(self, device, for_training: bool = False) -> Dict[str, torch.Tensor]:
        """for_training: only what the training forward / backward read (no BatchNorm folds, no dense position bias: train mode
        normalises with batch statistics and evaluates the bias MLP under autograd)."""
        ver = sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers())
        key = (ver, bool(for_training))
        if self._packed is not None and self._packed_version == key and self._packed_device == device:
            return self._packed
        C_, CP = self.embed_dim, _rup(self.embed_dim, 64)
        hid = int(C_ * self.expansion_factor)
        half = hid // 2
        HPh = _rup(half, 64)
        P: Dict[str, torch.Tensor] = {}

        def bn_fold(bn):
            s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
            return s, (bn.bias - bn.running_mean * s).float()

        with torch.no_grad(), batched_pack() as pk:
            for li, layer in enumerate(self.layers):
                nH = self.heads[li]
                dh, CA = C_ // nH, nH * 32
                hm = _head_map(nH, dh, device)
                qkv_rows = _qkv_rows(nH, dh, device)
                for bi, blk in enumerate(layer.blocks):
                    pre = f"{li}.{bi}."
                    at = blk.attn
                    P[pre + "Wqkv"] = _pack_linear(at.qkv.weight, 3 * CA, CP, row_map=qkv_rows)
                    P[pre + "bqkv"] = _pack_vec(at.qkv.bias, 3 * CA, row_map=qkv_rows, device=device)
                    P[pre + "Wproj"] = _pack_linear(at.proj.weight, CP, CA, col_map=hm)
                    P[pre + "bproj"] = _pack_vec(at.proj.bias, CP)
                    # DW-conv branch: conv bias + BatchNorm(eval) folded into scale / shift, channels scattered to the head-padded layout
                    w9 = torch.zeros(CA, 9, device=device)
                    w9[hm] = at.dwconv[0].weight.float().reshape(C_, 9)
                    P[pre + "dw_w"] = w9.contiguous()
                    if for_training:
                        self._pack_sgfn(P, pre, blk, hid, half, HPh, device)
                        continue
                    s, t = bn_fold(at.dwconv[1])
                    P[pre + "dw_s"] = _pack_vec(s, CA, row_map=hm)
                    P[pre + "dw_t"] = _pack_vec(t + at.dwconv[0].bias.float() * s, CA, row_map=hm)
                    ci = at.channel_interaction
                    s, t = bn_fold(ci[2])
                    S1 = ci[1].weight.shape[0]
                    w1 = torch.zeros(S1, CA, device=device)
                    w1[:, hm] = ci[1].weight.float().reshape(S1, C_) * s[:, None]
                    P[pre + "ci_w1"] = w1.contiguous()
                    P[pre + "ci_b1"] = (ci[1].bias.float() * s + t).contiguous()
                    w2 = torch.zeros(CA, S1, device=device)
                    w2[hm] = ci[4].weight.float().reshape(C_, S1)
                    P[pre + "ci_w2"] = w2.contiguous()
                    P[pre + "ci_b2"] = _pack_vec(ci[4].bias, CA, row_map=hm)
                    si = at.spatial_interaction
                    s, t = bn_fold(si[1])
                    S2 = si[0].weight.shape[0]
                    w0 = torch.zeros(S2, CA, device=device)
                    w0[:, hm] = si[0].weight.float().reshape(S2, C_) * s[:, None]
                    P[pre + "si_w0"] = w0.contiguous()
                    P[pre + "si_b0"] = (si[0].bias.float() * s + t).contiguous()
                    P[pre + "si_w3"] = si[3].weight.float().reshape(S2).contiguous()
                    P[pre + "si_b3"] = si[3].bias.float().reshape(1).contiguous()          # read by the kernel from device memory
                    if bi % 2 == 0:
                        for br, sa in enumerate(at.attns):      # dense bias of the branch: pos MLP on the offsets, gathered by the index
                            pos = sa.pos(sa.rpe_biases.float())
                            N = sa.H_sp * sa.W_sp
                            P[pre + f"bias{br}"] = pos[sa.relative_position_index.reshape(-1)].reshape(N, N, -1).permute(2, 0, 1).float().contiguous()
                    else:
                        P[pre + "temp"] = at.temperature.float().reshape(-1).contiguous()
                    self._pack_sgfn(P, pre, blk, hid, half, HPh, device)
                P[f"{li}.Wconv"] = _pack_conv(layer.conv.weight, CP, CP)
                P[f"{li}.bconv"] = _pack_vec(layer.conv.bias, CP)
            P["Wcab"] = _pack_conv(self.conv_after_body.weight, CP, CP)
            P["bcab"] = _pack_vec(self.conv_after_body.bias, CP)
            if self.upsampler == 'pixelshuffle':
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
            else:
                P["Wdirect"] = _pack_conv(self.upsample[0].weight, 16, CP)
                P["bdirect"] = _pack_vec(self.upsample[0].bias, 16)
            pk.resolve(P)
        self._packed, self._packed_version, self._packed_device = P, key, device
        return P

    def _pack_sgfn(self, P, pre, blk, hid, half, HPh, device):
        """SGFN: the two halves of the hidden vector each padded to HPh columns"""
        CP = _rup(self.embed_dim, 64)
        f = blk.ffn
        rows = _fc1_rows(hid, half, HPh, device)
        P[pre + "W1"] = _pack_linear(f.fc1.weight, 2 * HPh, CP, row_map=rows)
        P[pre + "b1"] = _pack_vec(f.fc1.bias, 2 * HPh, row_map=rows)
        P[pre + "W2"] = _pack_linear(f.fc2.weight, CP, HPh)
        P[pre + "b2"] = _pack_vec(f.fc2.bias, CP)
        sg9 = torch.zeros(HPh, 9, device=device)
        sg9[:half] = f.sg.conv.weight.float().reshape(half, 9)
        P[pre + "sg_w"] = sg9.contiguous()
        P[pre + "sg_s"] = _pack_vec(torch.ones(half, device=device), HPh)
        P[pre + "sg_t"] = _pack_vec(f.sg.conv.bias, HPh)

    # -- forward -----------------------------------------------------------------------------------------------------------
    def forward_features(self, x):
        raise NotImplementedError("forward_features is part of DAT.forward on the HIP path")

    def draw_drop_path(self, B: int, device) -> Optional[torch.Tensor]:
        """DropPath factors [n_blocks][2][B] (0 or 1 / keep, dat_arch.py:562-563 draws them per sample) or None when every rate is 0"""
        probs = [blk.drop_path_prob for layer in self.layers for blk in layer.blocks]
        if not any(pr > 0 for pr in probs):
            return None
        keep = getattr(self, "_keep_cache", None)          # built once per device: a host -> device upload cannot be graph-captured
        if keep is None or keep.device != torch.device(device):
            keep = self._keep_cache = 1.0 - torch.tensor(probs, dtype=torch.float32, device=device).view(-1, 1, 1)
        return (torch.rand(len(probs), 2, B, device=device) < keep).float() / keep

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("this DAT runs on MI355X through libsrk only; move the model and input to the GPU (no CPU fallback)")
        why = self._unsupported_reason()
        if why:
            raise SrkUnsupported(f"the MI355X HIP path does not cover {why}; no fallback path exists in this package")
        p0 = next(self.parameters())
        if p0.device != x.device:
            raise RuntimeError(f"input is on {x.device} but the model is on {p0.device}")
        _lib.claim_device(x.device.index if x.device.index is not None else torch.cuda.current_device())
        if self.training:
            # train mode: BatchNorm normalises with the batch's statistics and moves its running estimates, DropPath factors are drawn
            # here (dat_arch.py:562-563) and passed to the kernels as data; with grad enabled the whole model is ONE autograd node whose
            # backward is dat_train.dat_backward
            from .dat_train import DATFunction, dat_forward_train, pack_train
            drop = getattr(self, "_drop_override", None)          # training.GraphedTrainStep draws the factors outside its graph
            if drop is None:
                drop = self.draw_drop_path(x.shape[0], x.device)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                return DATFunction.apply(self, x, drop, *[p for _, p in self.named_parameters()])
            with torch.no_grad(), torch.cuda.device(x.device):
                return dat_forward_train(self, x.contiguous().float(), self._pack(x.device, True), pack_train(self, x.device), drop)[0]
        with torch.no_grad(), torch.cuda.device(x.device):
            return _dat_forward(self, x.contiguous().float(), self._pack(x.device))


def _dat_forward(m: DAT, x: torch.Tensor, P: Dict[str, torch.Tensor]) -> torch.Tensor:
    dev = x.device
    st = torch.cuda.current_stream(dev).cuda_stream
    B, Cin, H, W = x.shape
    s0, s1 = m.split_size
    big = max(s0, s1)
    Hp, Wp = _rup(H, big), _rup(W, big)        # window frame: q / k / v zero-padded to a multiple of the larger split (:376-384)
    T, HW, s = B * H * W, H * W, m.upscale
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    half = int(C_ * m.expansion_factor) // 2
    HPh = _rup(half, 64)
    f32, b16 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
    L = lib()
    mean3 = (C.c_float * 3)(*(m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]))
    img4 = torch.empty(T, 4, **f32)
    check(L.srk_img_prep(x.data_ptr(), img4.data_ptr(), B, Cin, H, W, H, W, float(m.img_range), C.byref(mean3), st))     # no padding in DAT.forward
    f0 = torch.empty(T, CP, **f32)
    check(L.srk_stem_conv(img4.data_ptr(), m.conv_first.weight.data_ptr(), m.conv_first.bias.data_ptr(), f0.data_ptr(), B, H, W, Cin, C_, CP, st))
    _, cur, _, _ = ops.layernorm_fwd(f0, m.before_RG[1].weight, m.before_RG[1].bias, C_, out_bf16=False, out_f32=True)

    CAmax = max(h * 32 for h in m.heads)
    qkv, att, conv, comb = (torch.empty(T, 3 * CAmax, **b16), torch.empty(T, CAmax, **b16), torch.empty(T, CAmax, **b16),
                            torch.empty(T, CAmax, **b16))
    cgate, tgate = torch.empty(B, CAmax, **f32), torch.empty(T, **f32)
    gate_ws = torch.empty(max(1, int(L.srk_channel_gate_workspace(B, HW, CAmax))), dtype=torch.uint8, device=dev)
    ca_ws = torch.empty(max(1, int(L.srk_channel_attention_workspace(B, HW, max(m.heads)))), dtype=torch.uint8, device=dev)
    xn2 = torch.empty(T, CP, **b16)
    hh = torch.empty(T, 2 * HPh, **b16)
    x2n, gated = torch.empty(T, HPh, **b16), torch.empty(T, HPh, **b16)
    xb = torch.empty(T, CP, **b16)
    stat_a, stat_b = torch.empty(T, **f32), torch.empty(T, **f32)
    stat_c, stat_d = torch.empty(T, **f32), torch.empty(T, **f32)
    ln_fusable = CP in (64, 128, 192)          # the LayerNorm that consumes a freshly written row rides in the producer's epilogue
    xn_a, xn_b = torch.empty(T, CP, **b16), torch.empty(T, CP, **b16)

    def next_norm(norm, dst):
        return dict(out=dst, mean=stat_c, rstd=stat_d, gamma=norm.weight, beta=norm.bias, C=C_) if ln_fusable else None

    xn1 = None            # norm1 of the upcoming block when the previous kernel already produced it in its epilogue
    for li, layer in enumerate(m.layers):
        nH = m.heads[li]
        CA, hb = nH * 32, nH // 2
        scale = float(m.qk_scale or (C_ // nH) ** -0.5)
        layer_in = cur
        nblk = len(layer.blocks)
        for bi, blk in enumerate(layer.blocks):
            pre = f"{li}.{bi}."
            if xn1 is None:
                xn1, _, _, _ = ops.layernorm_fwd(cur, blk.norm1.weight, blk.norm1.bias, C_)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
            # DW-conv branch on v (:418, :508): conv + BatchNorm + GELU
            check(L.srk_dwconv3x3(qkv.data_ptr() + 2 * CA * 2, 3 * CA, P[pre + "dw_w"].data_ptr(), P[pre + "dw_s"].data_ptr(), P[pre + "dw_t"].data_ptr(),
                                  None, 0, conv.data_ptr(), CA, B, H, W, CA // 8, 1, st))
            if bi % 2 == 0:
                at = blk.attn
                for br, (hs, wsz) in enumerate(((s0, s1), (s1, s0))):       # two window orientations on the two halves of the heads
                    sy, sx = (hs // 2, wsz // 2) if at.shifted else (0, 0)
                    off = br * hb * 32 * 2
                    check(L.srk_win_attention_fwd_padded(qkv.data_ptr() + off, 3 * CA, CA, P[pre + f"bias{br}"].data_ptr(), 0,
                                                         att.data_ptr() + off, CA, B, H, W, Hp, Wp, hs, wsz, sy, sx, hb, scale, 0, st))
                gate_src, tok_src, tok_on_a = conv, att, 0            # channel map from the conv branch, spatial map from the attention
            else:
                check(L.srk_channel_attention_fwd(qkv.data_ptr(), 3 * CA, CA, P[pre + "temp"].data_ptr(), ca_ws.data_ptr(), att.data_ptr(), CA, B, HW,
                                                  nH, C_ // nH, st))
                gate_src, tok_src, tok_on_a = att, conv, 1            # channel map from the attention, spatial map from the conv branch
            S1, S2 = P[pre + "ci_w1"].shape[0], P[pre + "si_w0"].shape[0]
            check(L.srk_channel_gate_act(gate_src.data_ptr(), gate_ws.data_ptr(), P[pre + "ci_w1"].data_ptr(), P[pre + "ci_b1"].data_ptr(),
                                         P[pre + "ci_w2"].data_ptr(), P[pre + "ci_b2"].data_ptr(), 1.0, cgate.data_ptr(), B, HW, CA, CA, S1, 1, st))
            check(L.srk_spatial_gate_dev(tok_src.data_ptr(), CA, P[pre + "si_w0"].data_ptr(), P[pre + "si_b0"].data_ptr(), P[pre + "si_w3"].data_ptr(),
                                         P[pre + "si_b3"].data_ptr(), S2, tgate.data_ptr(), T, CA, st))
            check(L.srk_dual_gate_combine(att.data_ptr(), conv.data_ptr(), cgate.data_ptr(), tgate.data_ptr(), comb.data_ptr(), T, HW, CA, tok_on_a, st))
            x1 = torch.empty(T, CP, **f32)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, comb, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1,
                  xn=dict(out=xn2, mean=stat_a, rstd=stat_b, gamma=blk.norm2.weight, beta=blk.norm2.bias, C=C_))
            # SGFN (:74-90): fc1 + GELU, x1 * DWconv(LN(x2)), fc2 + residual
            _gemm(st, _lib.LD_ROWS, _lib.EP_GELU, xn2, P[pre + "W1"], T, 2 * HPh, CP, lda=CP, bias=P[pre + "b1"], outb2=hh)
            check(L.srk_rowln_bf16(hh.data_ptr() + HPh * 2, 2 * HPh, blk.ffn.sg.norm.weight.data_ptr(), blk.ffn.sg.norm.bias.data_ptr(), x2n.data_ptr(),
                                   HPh, T, half, HPh, st))
            check(L.srk_dwconv3x3(x2n.data_ptr(), HPh, P[pre + "sg_w"].data_ptr(), P[pre + "sg_s"].data_ptr(), P[pre + "sg_t"].data_ptr(), hh.data_ptr(),
                                  2 * HPh, gated.data_ptr(), HPh, B, H, W, HPh // 8, 0, st))
            nxt = torch.empty(T, CP, **f32)
            last = bi == nblk - 1
            dst = xn_a if xn1 is not xn_a else xn_b
            nn_ = None if last else next_norm(layer.blocks[bi + 1].norm1, dst)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, gated, P[pre + "W2"], T, CP, HPh, lda=HPh, bias=P[pre + "b2"], res=x1, outf=nxt,
                  outb=xb if last else None, xn=nn_)
            cur = nxt
            xn1 = dst if nn_ is not None else None
        nxt = torch.empty(T, CP, **f32)                               # ResidualGroup :653-657: conv + residual
        nn_ = next_norm(m.layers[li + 1].blocks[0].norm1 if li + 1 < len(m.layers) else m.norm, xn_a)        # next norm1 / the final norm :750
        _gemm(st, _lib.LD_CONV3, _lib.EP_RES, xb, P[f"{li}.Wconv"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P[f"{li}.bconv"], res=layer_in, outf=nxt,
              xn=nn_)
        cur = nxt
        xn1 = xn_a if nn_ is not None else None

    xnf = xn1 if xn1 is not None else ops.layernorm_fwd(cur, m.norm.weight, m.norm.bias, C_)[0]
    fb = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_RES_BF16, xnf, P["Wcab"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P["bcab"], res=f0, outb=fb)
    y = torch.empty(B, Cin, H * s, W * s, **f32)
    mean4 = (m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]) + [0.0]
    img = dict(inv_range=1.0 / float(m.img_range), Cimg=Cin, Hc=H * s, Wc=W * s, mean=mean4)
    if m.upsampler == 'pixelshuffle':
        t1 = torch.empty(T, 64, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_LRELU, fb, P["Wbefore"], T, 64, 9 * CP, conv=(B, H, W, CP), bias=P["bbefore"], outb=t1, scale=0.01)
        src, h, w, k = t1, H, W, 0
        while f"Wup{k}" in P:
            r = int(P[f"rup{k}"])
            N = P[f"Wup{k}"].shape[0]
            up = torch.empty(B * h * r * w * r, 64, **b16)
            _gemm(st, _lib.LD_CONV3, _lib.EP_PS, src, P[f"Wup{k}"], B * h * w, N, 9 * 64, conv=(B, h, w, 64), bias=P[f"bup{k}"], outb=up, r=r, Cs=64, ldo=N)
            src, h, w, k = up, h * r, w * r, k + 1
        _gemm(st, _lib.LD_CONV3, _lib.EP_IMG, src, P["Wlast"], B * h * w, 16, 9 * 64, conv=(B, h, w, 64), bias=P["blast"], outf=y, img=img)
    else:
        a = dict(img)
        _gemm(st, _lib.LD_CONV3, _lib.EP_PS_IMG, fb, P["Wdirect"], T, 16, 9 * CP, conv=(B, H, W, CP), bias=P["bdirect"], outf=y, img=a, r=s)
    return y
