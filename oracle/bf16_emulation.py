"""bf16-rounding emulation of the HIP pipeline on the CPU -- TEST INFRASTRUCTURE ONLY.

``swinir_oracle.py`` is the fp32 restatement of the reference; the HIP path differs from it by bf16
rounding noise (0.3-0.6 % forward, 1-7 % per gradient tensor), which is too loose to prove the
*logic* of the fused kernels (index maps, LayerNorm-backward algebra, DropPath scaling, attention
backward, pixel-shuffle gradient routing ...).  This module restates the same forward in fp32 torch but
rounds to bf16 at exactly the points where the HIP pipeline stores or feeds bf16,
in forward AND backward (gradient-rounding hooks), so that HIP and emulation agree up to accumulation
order (~1e-3), independent of the bf16 noise floor.

Rounding points (forward): packed weights; LayerNorm outputs; q*scale, k, v; softmax probabilities fed to
P.V; attention output; MLP pre-activation u (stored) -- GELU is applied to the fp32 u; hidden h; the bf16
copies that feed the convs (last block output, final norm output, conv_after_body + skip, LeakyReLU
output, pixel-shuffled activations).  Backward: every gradient operand of a dgrad / wgrad GEMM is bf16
(dy of each linear/conv), the attention backward rounds P and dS before their products, LayerNorm backward
and the residual gradient stream are fp32.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import swinir_oracle as O

Tensor = torch.Tensor


def _bf(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _Rnd(torch.autograd.Function):
    """value rounded to bf16, gradient passed through."""
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _GRnd(torch.autograd.Function):
    """identity forward, gradient rounded to bf16."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


rnd = _Rnd.apply
grnd = _GRnd.apply


def qlinear(x_q: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """y = x_q . bf16(w)^T + b with fp32 accumulation; dy is rounded to bf16 before dgrad / wgrad."""
    return grnd(F.linear(x_q, rnd(w), b))


def qconv(x_q: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    return grnd(F.conv2d(x_q, rnd(w), b, padding=1))


class _WindowAttention(torch.autograd.Function):
    """softmax(q k^T + bias + mask) v with the HIP kernels' rounding points (attn.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, bias, mask):
        # q, k, v: bf16-valued fp32 [B_, nH, 64, d]; bias [nH,64,64]; mask [nW,64,64] or None
        s = q @ k.transpose(-1, -2) + bias[None]
        if mask is not None:
            B_, nH = q.shape[:2]
            nW = mask.shape[0]
            s = (s.view(B_ // nW, nW, nH, 64, 64) + mask[None, :, None]).view(B_, nH, 64, 64)
        p = torch.softmax(s, dim=-1)
        ctx.save_for_backward(q, k, v, p)
        # forward kernels feed the UNNORMALISED numerators exp(s - max) to the P.V MFMA as bf16 and scale the outputs
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        return _bf((_bf(e) @ v) / e.sum(dim=-1, keepdim=True))

    @staticmethod
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        do = _bf(do)
        dp = do @ v.transpose(-1, -2)
        ds = p * (dp - (p * dp).sum(-1, keepdim=True))
        dv = _bf(p).transpose(-1, -2) @ do
        dsq = _bf(ds)
        dk = dsq.transpose(-1, -2) @ q
        dq = dsq @ k
        return dq, dk, dv, ds.sum(0), None


def emul_block(x: Tensor, H: int, W: int, sd: Dict[str, Tensor], pre: str, nH: int, shift: int,
               f_attn: Optional[Tensor], f_mlp: Optional[Tensor], qk_scale: Optional[float]):
    """One Swin block; returns (x2, h-branch info is internal).  f_* are per-sample DropPath factors [B] or None."""
    B, L, C = x.shape
    d = C // nH
    scale = qk_scale or d ** -0.5
    idx = torch.from_numpy(O.window_token_index(H, W, 8, shift)).reshape(-1)
    nW = idx.numel() // 64
    fa = None if f_attn is None else f_attn.view(B, 1, 1)
    fm = None if f_mlp is None else f_mlp.view(B, 1, 1)

    xn1 = rnd(F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5))
    xw = xn1[:, idx].reshape(B * nW, 64, C)
    qkv = qlinear(xw, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"]).reshape(B * nW, 64, 3, nH, d)
    q = rnd(qkv[:, :, 0].transpose(1, 2) * scale)
    k = rnd(qkv[:, :, 1].transpose(1, 2))
    v = rnd(qkv[:, :, 2].transpose(1, 2))
    bias = O.dense_rel_pos_bias(sd[pre + "attn.relative_position_bias_table"], 8)
    mask = torch.from_numpy(O.shift_attn_mask(H, W, 8, shift)) if shift > 0 else None
    ao = _WindowAttention.apply(q, k, v, bias, mask)                    # bf16-valued [B_, nH, 64, d]
    ao = ao.transpose(1, 2).reshape(B * nW, 64, C)
    # proj: its dy operand is the bf16 copy of the x1-gradient in window order, scaled by the attention DropPath factor
    pr = F.linear(ao, rnd(sd[pre + "attn.proj.weight"]), sd[pre + "attn.proj.bias"]).reshape(B, nW * 64, C)
    a = torch.empty_like(x)
    a[:, idx] = pr
    a = grnd(a)            # the proj GEMMs see dy = bf16(f_attn * d x1) (gxbw)
    x1 = x + (a * fa if fa is not None else a)

    xn2 = rnd(F.layer_norm(x1, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5))
    u = F.linear(xn2, rnd(sd[pre + "mlp.fc1.weight"]), sd[pre + "mlp.fc1.bias"])
    u = grnd(u)                                                         # du is stored bf16
    h = rnd(_GeluFromStoredU.apply(u))
    m = F.linear(h, rnd(sd[pre + "mlp.fc2.weight"]), sd[pre + "mlp.fc2.bias"])
    m = grnd(m)            # the fc2 GEMMs see dy = bf16(f_mlp * d x2) (gxb2)
    return x1 + (m * fm if fm is not None else m)


class _GeluFromStoredU(torch.autograd.Function):
    """h = gelu(u) on the fp32 u; the derivative is evaluated at the STORED bf16 u (EP_DGELU reads aux = bf16 u)."""
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(_bf(u))
        return F.gelu(u)

    @staticmethod
    def backward(ctx, g):
        (ub,) = ctx.saved_tensors
        cdf = 0.5 * (1.0 + torch.erf(ub * 0.70710678118654752))
        pdf = 0.39894228040143268 * torch.exp(-0.5 * ub * ub)
        return g * (cdf + ub * pdf)


def resi_conv_emul(x_q: Tensor, sd: Dict[str, Tensor], name: str, kind: str) -> Tensor:
    """'1conv' / '3conv' residual-connection conv as the executor runs it (csrc/swinir.hip resi_forward): every conv reads a
    bf16 activation and bf16 weights; the two narrow LeakyReLU(0.2) outputs are stored in bf16."""
    if kind == "1conv":
        return qconv(x_q, sd[name + ".weight"], sd[name + ".bias"])
    a1 = rnd(F.leaky_relu(qconv(x_q, sd[name + ".0.weight"], sd[name + ".0.bias"]), 0.2))
    a2 = rnd(F.leaky_relu(grnd(F.conv2d(a1, rnd(sd[name + ".2.weight"]), sd[name + ".2.bias"])), 0.2))
    return qconv(a2, sd[name + ".4.weight"], sd[name + ".4.bias"])


def swinir_forward_emul(sd: Dict[str, Tensor], cfg: O.SwinIRConfig, x: Tensor, drop_keep: Optional[Tensor] = None) -> Tensor:
    """HIP-pipeline emulation of SwinIR.forward for the heads the HIP path covers."""
    assert cfg.window_size == 8
    H0, W0 = x.shape[2:]
    ph, pw = (8 - H0 % 8) % 8, (8 - W0 % 8) % 8
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph), mode="reflect")
    B, _, H, W = x.shape
    mean = torch.tensor([0.4488, 0.4371, 0.4040]).view(1, 3, 1, 1) if cfg.in_chans == 3 else torch.zeros(1, 1, 1, 1)
    x = (x - mean) * cfg.img_range
    C = cfg.embed_dim
    f0 = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)          # fp32 stem
    t = f0.flatten(2).transpose(1, 2)
    # head LayerNorm backward consumes the bf16 copy of the gradient stream
    t = grnd(F.layer_norm(t, (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], 1e-5))
    res0 = cfg.img_size // cfg.patch_size
    blk = 0
    for li, depth in enumerate(cfg.depths):
        y = t
        for bi in range(depth):
            ws, shift = O.effective_window(H, W, 8, 0 if bi % 2 == 0 else 4, (res0, res0))
            fa = None if drop_keep is None else drop_keep[blk, 0]
            fm = None if drop_keep is None else drop_keep[blk, 1]
            y = emul_block(y, H, W, sd, f"layers.{li}.residual_group.blocks.{bi}.", cfg.num_heads[li], shift, fa, fm, cfg.qk_scale)
            blk += 1
        yb = rnd(y).transpose(1, 2).reshape(B, C, H, W)                                  # bf16 copy feeds the RSTB conv
        cv = resi_conv_emul(yb, sd, f"layers.{li}.conv", cfg.resi_connection)
        t = cv.flatten(2).transpose(1, 2) + t
    # the final LayerNorm's backward reads a bf16 dy (dxn)
    xn = rnd(grnd(F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"], 1e-5))).transpose(1, 2).reshape(B, C, H, W)
    fb = rnd(resi_conv_emul(xn, sd, "conv_after_body", cfg.resi_connection) + grnd(f0))
    s = cfg.upscale
    if cfg.upsampler == "pixelshuffle":
        f = rnd(F.leaky_relu(qconv(fb, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"]), 0.01))
        stages = int(math.log2(s)) if s & (s - 1) == 0 else 1
        r = 2 if s & (s - 1) == 0 else 3
        for i in range(stages):
            f = rnd(O.pixel_shuffle(qconv(f, sd[f"upsample.{2 * i}.weight"], sd[f"upsample.{2 * i}.bias"]), r))
        out = _ConvLast.apply(f, sd["conv_last.weight"], sd["conv_last.bias"])
    elif cfg.upsampler == "pixelshuffledirect":
        out = O.pixel_shuffle(_ConvLast.apply(fb, sd["upsample.0.weight"], sd["upsample.0.bias"]), s)
    elif cfg.upsampler == "nearest+conv":
        f = rnd(F.leaky_relu(qconv(fb, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"]), 0.01))
        for name in (("conv_up1", "conv_up2") if s == 4 else ("conv_up1",)):
            # the 2x2 sum of the upsample's backward is formed in fp32 and rounded once, with the LeakyReLU factor applied
            f = rnd(F.leaky_relu(qconv(grnd(F.interpolate(f, scale_factor=2, mode="nearest")), sd[name + ".weight"],
                                       sd[name + ".bias"]), 0.2))
        f = rnd(F.leaky_relu(qconv(f, sd["conv_hr.weight"], sd["conv_hr.bias"]), 0.2))
        out = _ConvLast.apply(f, sd["conv_last.weight"], sd["conv_last.bias"])
    else:
        out = x + _ConvLast.apply(fb, sd["conv_last.weight"], sd["conv_last.bias"])
    out = out / cfg.img_range + mean
    return out[:, :, :H0 * s, :W0 * s]


class _ConvLast(torch.autograd.Function):
    """Image-head conv: forward with bf16 packed weights (MFMA path); dgrad with the fp32 weights and a bf16 result,
    wgrad in fp32 from the stored bf16 activations (small-Cout VALU kernels, misc.hip)."""
    @staticmethod
    def forward(ctx, x_q, w, b):
        ctx.save_for_backward(x_q, w)
        return F.conv2d(x_q, _bf(w), b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x_q, w = ctx.saved_tensors
        dx = _bf(torch.nn.grad.conv2d_input(x_q.shape, w, g, padding=1))
        dw = torch.nn.grad.conv2d_weight(x_q, w.shape, g, padding=1)
        return dx, dw, g.sum((0, 2, 3))


def loss_and_grads_emul(sd: Dict[str, Tensor], cfg: O.SwinIRConfig, lr_img: Tensor, hr_img: Tensor,
                        drop_keep: Optional[Tensor] = None):
    keys = O.param_keys(cfg)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    full = dict(sd)
    full.update(leaves)
    out = swinir_forward_emul(full, cfg, lr_img, drop_keep)
    loss = O.l1_loss(out, hr_img)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys])
    return loss.detach(), out.detach(), dict(zip(keys, grads))
