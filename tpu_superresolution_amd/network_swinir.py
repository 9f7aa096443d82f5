"""SwinIR with the reference's constructor, module tree and state_dict, executed by libsrk on MI355X.

Drop-in for ``modules/network_swinir.py`` of ViacheslavTimofeev/tpu_superresolution: same
``SwinIR(...)`` keyword arguments (network_swinir.py:646-652), same parameter / buffer names and
shapes (so the public SwinIR ``.pth`` files and the reference's own checkpoints load with
``strict=True``), same ``forward(x[B,C,H,W]) -> [B,C,H*s,W*s]``.  The module tree below only *holds*
the parameters; the arithmetic of ``SwinIR.forward`` and its backward run in hand-written HIP
kernels behind the C ABI of ``include/srk.h``.  There is no CPU path: a CPU tensor, a missing
``libsrk.so`` or a configuration the kernels do not cover raises.

Covered by the HIP path: window_size 8, head_dim <= 32, embed_dim <= 256, in_chans 1/3, resi_connection '1conv' and
'3conv', patch_norm=True, ape (at img_size), use_checkpoint, all four heads -- 'pixelshuffle' (x2/x3/x4/x8), 'pixelshuffledirect'
(upscale^2 * in_chans <= 16), 'nearest+conv' (x2/x4) and '' (denoising, upscale 1) -- and ``forward_features`` as a callable
(inference).  window_size 16 runs INFERENCE through the 256-token window attention of the HAT path (``swinir_w16.py``; the two
pixel-shuffle heads, '1conv').  Other constructor options (other window sizes, patch_norm=False, dropout > 0) build the same
state_dict but raise ``NotImplementedError`` in ``forward``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import SrkUnsupported
from .engine import SwinIREngine, SwinIRPlan, _SwinIRFunction


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def window_partition(x, window_size):
    """(B, H, W, C) -> (num_windows*B, window_size, window_size, C)   [network_swinir.py:33-45]"""
    return ops.window_partition(x, window_size)


def window_reverse(windows, window_size, H, W):
    """(num_windows*B, window_size, window_size, C) -> (B, H, W, C)   [network_swinir.py:48-62]"""
    return ops.window_reverse(windows, window_size, H, W)


def _holder_forward(self, *a, **k):
    raise NotImplementedError(f"{type(self).__name__} only holds parameters here; run the enclosing SwinIR.forward "
                              "(the whole model executes inside libsrk)")


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    forward = _holder_forward


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        ys, xs = torch.arange(wh).repeat_interleave(ww), torch.arange(ww).repeat(wh)
        rpi = (ys[:, None] - ys[None, :] + wh - 1) * (2 * ww - 1) + (xs[:, None] - xs[None, :] + ww - 1)
        self.register_buffer("relative_position_index", rpi)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    forward = _holder_forward

    def extra_repr(self):
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"


def _shift_mask_cpu(H, W, ws, shift):
    lab_h = torch.where(torch.arange(H) < H - ws, 0, torch.where(torch.arange(H) < H - shift, 1, 2))
    lab_w = torch.where(torch.arange(W) < W - ws, 0, torch.where(torch.arange(W) < W - shift, 1, 2))
    lab = (lab_h[:, None] * 3 + lab_w[None, :]).view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = lab[:, None, :] != lab[:, :, None]
    return torch.where(diff, torch.tensor(-100.0), torch.tensor(0.0))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(self.input_resolution) <= self.window_size:     # network_swinir.py:193-196
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.drop_path_prob = float(drop_path)
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=_pair(self.window_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        mask = self.calculate_mask(self.input_resolution) if self.shift_size > 0 else None
        self.register_buffer("attn_mask", mask)

    def calculate_mask(self, x_size):
        """[nW, N, N] in {0, -100}  (network_swinir.py:216-237).  Kept for state_dict compatibility; the HIP
        attention kernel derives the same mask arithmetically from the window position."""
        H, W = x_size
        return _shift_mask_cpu(H, W, self.window_size, self.shift_size)

    forward = _holder_forward

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if i % 2 == 0 else window_size // 2, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                 qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])
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


class RSTB(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 img_size=224, patch_size=4, resi_connection='1conv'):
        super().__init__()
        self.dim, self.input_resolution = dim, input_resolution
        self.residual_group = BasicLayer(dim=dim, input_resolution=input_resolution, depth=depth, num_heads=num_heads,
                                         window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                         drop=drop, attn_drop=attn_drop, drop_path=drop_path, norm_layer=norm_layer,
                                         downsample=downsample, use_checkpoint=use_checkpoint)
        self.conv = _resi_conv(dim, resi_connection)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=dim, norm_layer=None)
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=0, embed_dim=dim)

    forward = _holder_forward


def _resi_conv(dim, kind):
    if kind == '1conv':
        return nn.Conv2d(dim, dim, 3, 1, 1)
    if kind == '3conv':
        return nn.Sequential(nn.Conv2d(dim, dim // 4, 3, 1, 1), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                             nn.Conv2d(dim // 4, dim // 4, 1, 1, 0), nn.LeakyReLU(negative_slope=0.2, inplace=True),
                             nn.Conv2d(dim // 4, dim, 3, 1, 1))
    raise ValueError(f"unknown resi_connection {kind!r}")


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


class UpsampleOneStep(nn.Sequential):
    def __init__(self, scale, num_feat, num_out_ch, input_resolution=None):
        self.num_feat, self.input_resolution = num_feat, input_resolution
        super().__init__(nn.Conv2d(num_feat, (scale ** 2) * num_out_ch, 3, 1, 1), nn.PixelShuffle(scale))


class SwinIR(nn.Module):
    r"""SwinIR (Liang et al. 2021) -- reference constructor signature, network_swinir.py:646-652."""

    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=[6, 6, 6, 6], num_heads=[6, 6, 6, 6],
                 window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True, use_checkpoint=False, upscale=2,
                 img_range=1., upsampler='', resi_connection='1conv', **kwargs):
        super().__init__()
        num_feat = 64
        self.img_range, self.upscale, self.upsampler, self.window_size = img_range, upscale, upsampler, window_size
        self.in_chans, self.embed_dim, self.num_features = in_chans, embed_dim, embed_dim
        self.depths, self.heads = list(depths), list(num_heads)
        self.num_layers, self.ape, self.patch_norm, self.mlp_ratio = len(depths), ape, patch_norm, mlp_ratio
        self.qkv_bias, self.qk_scale, self.resi_connection, self.patch_size = qkv_bias, qk_scale, resi_connection, patch_size
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        self.use_checkpoint = bool(use_checkpoint)       # recompute policy of the training executor (include/srk.h: use_checkpoint)
        self.mean = torch.Tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1) if in_chans == 3 else torch.zeros(1, 1, 1, 1)

        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        self.patches_resolution = self.patch_embed.patches_resolution
        self.patch_unembed = PatchUnEmbed(img_size=img_size, patch_size=patch_size, in_chans=embed_dim, embed_dim=embed_dim)
        if ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]     # network_swinir.py:701
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(RSTB(dim=embed_dim, input_resolution=tuple(self.patches_resolution), depth=depths[i],
                                    num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                                    drop_path=self.dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                                    downsample=None, use_checkpoint=use_checkpoint, img_size=img_size, patch_size=patch_size,
                                    resi_connection=resi_connection))
        self.norm = norm_layer(self.num_features)
        self.conv_after_body = _resi_conv(embed_dim, resi_connection)
        if upsampler == 'pixelshuffle':
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = Upsample(upscale, num_feat)
            self.conv_last = nn.Conv2d(num_feat, in_chans, 3, 1, 1)
        elif upsampler == 'pixelshuffledirect':
            self.upsample = UpsampleOneStep(upscale, embed_dim, in_chans, tuple(self.patches_resolution))
        elif upsampler == 'nearest+conv':
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.conv_up1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
            if upscale == 4:
                self.conv_up2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_last = nn.Conv2d(num_feat, in_chans, 3, 1, 1)
            self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        else:
            self.conv_last = nn.Conv2d(embed_dim, in_chans, 3, 1, 1)
        self.apply(self._init_weights)

        self._plan: Optional[SwinIRPlan] = None
        self._engine: Optional[SwinIREngine] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_params_dirty())

    # -- reference helper API ---------------------------------------------------------------------------
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def check_image_size(self, x):
        """Reflect-pad H, W up to a multiple of window_size (network_swinir.py:783-788).  The executor folds this
        into its image-preparation kernel; this method is kept for API compatibility."""
        _, _, h, w = x.size()
        ph = (self.window_size - h % self.window_size) % self.window_size
        pw = (self.window_size - w % self.window_size) % self.window_size
        return torch.nn.functional.pad(x, (0, pw, 0, ph), 'reflect')

    def flops(self):
        """Forward MACs*1 in the reference's own convention (network_swinir.py:842-851), made to work for every
        upsampler (the reference raises AttributeError for 'pixelshuffle')."""
        H, W = self.patches_resolution
        C, N = self.embed_dim, self.window_size * self.window_size
        f = H * W * 3 * C * 9 + H * W * C
        hid = int(C * self.mlp_ratio)
        for d in self.depths:
            per_blk = 2 * C * H * W + (H * W // N) * (N * C * 3 * C + 2 * N * N * C + N * C * C) + 2 * H * W * C * hid
            f += d * per_blk + H * W * C * C * 9
        f += H * W * 3 * C * C
        if self.upsampler == 'pixelshuffledirect':
            f += H * W * C * 3 * 9
        return f

    # -- engine plumbing ----------------------------------------------------------------------------------
    def mark_params_dirty(self):
        """Call after changing parameters behind the module's back (in-place edits of ``p.data`` in eval mode)."""
        if getattr(self, "_engine", None) is not None:
            self._engine.packed_valid = False

    def train(self, mode: bool = True):
        if mode != self.training:
            self.mark_params_dirty()      # a train -> eval switch usually follows optimizer steps
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._engine = None        # .to()/.cuda()/.float() re-create the parameter tensors: re-bind lazily
        return super()._apply(fn, *args, **kwargs)

    def _unsupported_reason(self) -> Optional[str]:
        if not self.patch_norm:
            return "patch_norm=False"
        if not self.qkv_bias:
            return "qkv_bias=False"
        if self.patch_size != 1:
            return f"patch_size={self.patch_size}"
        if self.drop_rate or self.attn_drop_rate:
            return "dropout > 0"
        return None

    def _bind(self, device: torch.device) -> SwinIREngine:
        if self._engine is not None and self._engine.device == device:
            return self._engine
        why = self._unsupported_reason()
        if why:
            raise SrkUnsupported(f"the MI355X HIP path does not cover {why}; no fallback path exists in this package")
        if self._plan is None:
            self._plan = SwinIRPlan(img_size=min(self.patches_resolution), in_chans=self.in_chans, embed_dim=self.embed_dim,
                                    depths=self.depths, num_heads=self.heads, window_size=self.window_size,
                                    mlp_ratio=self.mlp_ratio, upscale=self.upscale, img_range=self.img_range,
                                    upsampler=self.upsampler, qk_scale=self.qk_scale, resi_connection=self.resi_connection,
                                    use_checkpoint=self.use_checkpoint, ape=bool(self.ape),
                                    options=getattr(self, "plan_options", None))      # per-model kernel options: set model.plan_options = {...} before the first forward
        eng = SwinIREngine(self._plan, device)
        named = dict(self.named_parameters())
        missing = [p.name for p in self._plan.params if p.name not in named]
        extra = [k for k in named if k not in {p.name for p in self._plan.params}]
        if missing or extra:
            raise RuntimeError(f"parameter table mismatch: missing={missing[:3]} extra={extra[:3]}")
        views = eng.views(eng.flat)
        with torch.no_grad():
            for name, v in views.items():
                p = named[name]
                if tuple(p.shape) != tuple(v.shape):
                    raise RuntimeError(f"shape mismatch for {name}: {tuple(p.shape)} vs {tuple(v.shape)}")
                v.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = v                     # parameters become views of the flat buffer
                p.grad = None
        self._engine = eng
        return eng

    def _param_views_ok(self, eng: SwinIREngine) -> bool:
        p0 = self.conv_first.weight
        return p0.data_ptr() == eng.flat.data_ptr() + 4 * eng.plan.params[0].offset

    def _drop_scale(self, B: int, device) -> Optional[torch.Tensor]:
        """Per-sample DropPath factors [n_blocks, 2, B] (timm DropPath semantics: Bernoulli(keep)/keep), drawn
        only in training mode with drop_path_rate > 0 (network_swinir.py:204, :276-277)."""
        if not self.training or self.drop_path_rate <= 0:
            return None
        keep = getattr(self, "_keep_cache", None)
        if keep is None or keep.device != torch.device(device):
            # built once per device: a per-step host->device copy is a synchronous upload and cannot be graph-captured
            keep = 1.0 - torch.tensor(self.dpr, dtype=torch.float32, device=device).view(-1, 1, 1)
            self._keep_cache = keep
        u = torch.rand((len(self.dpr), 2, B), dtype=torch.float32, device=device)
        return ((u < keep).float() / keep).contiguous()

    def _backward_into_flat(self, d_y, shape, drop_scale):
        eng = self._engine
        fresh = all(p.grad is None for p in self.parameters())
        g = eng.ensure_grad()
        if fresh:
            g.zero_()
        eng.backward(d_y, shape, drop_scale)
        gv = eng.views(g)
        for name, p in self.named_parameters():
            if p.requires_grad and p.grad is None:
                p.grad = gv[name]

    # -- forward ----------------------------------------------------------------------------------------------
    def forward_features(self, x):
        """network_swinir.py:790-803 as a callable: x = conv_first's output [B, embed_dim, H, W] (H, W multiples of the
        window size) -> patch_embed norm -> RSTBs -> norm -> [B, embed_dim, H, W].  ``forward`` does not call this (the
        executor runs the same kernels inside one fused sequence); it exists for code that uses the method directly.
        Inference only: no autograd graph is recorded."""
        if not x.is_cuda:
            raise RuntimeError("this SwinIR runs on MI355X through libsrk only (no CPU fallback exists in this package)")
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
            raise RuntimeError("forward_features on the HIP path is inference-only; call it under torch.no_grad() / in eval mode "
                               "(training goes through SwinIR.forward)")
        eng = self._bind(x.device)
        if not self._param_views_ok(eng):
            self._engine = None
            eng = self._bind(x.device)
        ver = sum(p._version for p in self.parameters())
        if ver != eng.pack_version:
            eng.packed_valid = False
            eng.pack_version = ver
        return eng.forward_features(x)

    def forward(self, x, drop_scale: Optional[torch.Tensor] = None):
        if not x.is_cuda:
            raise RuntimeError("this SwinIR runs on MI355X through libsrk only; move the model and input to the GPU "
                               "(no CPU fallback exists in this package)")
        if self.window_size == 16:        # 256-token windows: host-orchestrated inference on the HAT path's attention kernel
            if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
                raise SrkUnsupported("SwinIR(window_size=16) on the HIP path is inference-only: call model.eval() / torch.no_grad() "
                                     "(training runs with window_size 8)")
            from . import swinir_w16
            return swinir_w16.forward(self, x)
        eng = self._bind(x.device)
        if not self._param_views_ok(eng):
            self._engine = None
            eng = self._bind(x.device)
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        needs_grad = torch.is_grad_enabled() and anchor is not None
        if self.training or needs_grad:
            eng.packed_valid = False          # parameters may have been stepped since the last call
            eng.pack_version = -1
        else:
            # inference reuses the bf16 pack only while no parameter has been written in place since it was made
            # (torch.optim.* steps, p.mul_(), ... bump the autograd version counter; FusedAdamW / load_state_dict /
            # mark_params_dirty / train()<->eval() switches clear packed_valid themselves)
            ver = sum(p._version for p in self.parameters())
            if ver != eng.pack_version:
                eng.packed_valid = False
                eng.pack_version = ver
        if drop_scale is None:
            drop_scale = self._drop_scale(x.shape[0], x.device)
        if needs_grad:
            return _SwinIRFunction.apply(x, anchor, self, drop_scale, True)
        return eng.forward(x, False, drop_scale)
