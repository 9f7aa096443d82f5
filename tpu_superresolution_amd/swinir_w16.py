"""SwinIR with window_size 16 on the HIP path (inference): the 256-token window attention of the HAT path applied to SwinIR's blocks.

The C++ executor (csrc/swinir.hip) is built around 64-token windows (window_size 8, every configuration the reference's scripts
use).  ``SwinIR(window_size=16)`` -- network_swinir.py builds any window size (:640-760) -- runs here instead, as a host-side
sequence of C-ABI calls in the manner of ``hat_arch._hat_forward``: a Swin block with 16 x 16 windows is HAT's HAB without its conv
branch (hat_arch.py:281-325 vs network_swinir.py:240-279), so the kernels are the same:

    check_image_size + normalise (:783-788, :807-809)     srk_img_prep (reflect padding to a multiple of 16)
    conv_first, patch_embed.norm                          srk_stem_conv, srk_layernorm_fwd
    norm1 -> qkv -> (S)W-MSA -> proj + shortcut -> norm2   srk_gemm_ex, srk_win256_attention_fwd (roll / partition / reverse in the addresses,
                                                          arithmetic shift mask, the 961-row bias table indexed in the kernel), next norm fused
    Mlp + shortcut                                        srk_mlp_fused_fwd (width 180) or fc1 + GELU / fc2 + residual GEMMs
    RSTB conv + skip, conv_after_body, head               implicit-GEMM 3x3 convs with residual / LeakyReLU / PixelShuffle / image epilogues

Inference only ('pixelshuffle' and 'pixelshuffledirect' heads, resi_connection '1conv', no ape); a grad-enabled training forward raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import SrkUnsupported, check, lib
from .hat_arch import _gemm, _head_map, _pack_conv, _pack_linear, _pack_vec, _ps_map, _ptr, _qkv_rows, _rup, batched_pack


def unsupported_reason(m) -> str:
    if m.upsampler not in ("pixelshuffle", "pixelshuffledirect"):
        return f"window_size=16 with upsampler={m.upsampler!r}"
    if m.resi_connection != "1conv" or m.ape or not m.patch_norm or not m.qkv_bias or m.patch_size != 1 or m.drop_rate or m.attn_drop_rate:
        return "window_size=16 with resi_connection != '1conv', ape, patch_norm=False, qkv_bias=False, patch_size != 1 or dropout"
    if m.embed_dim > 256 or any(m.embed_dim % h or m.embed_dim // h > 32 for h in m.heads):
        return "embed_dim > 256 or head_dim > 32"
    if m.upsampler == "pixelshuffledirect" and m.upscale ** 2 * m.in_chans > 16:
        return "pixelshuffledirect with upscale^2 * in_chans > 16"
    if any(blk.window_size != 16 for layer in m.layers for blk in layer.residual_group.blocks):
        return "window_size=16 with img_size <= 16 (the blocks fall back to one window of the image size)"
    return ""


def pack(m, device) -> Dict[str, torch.Tensor]:
    ver = sum(p._version for p in m.parameters())
    if getattr(m, "_w16_packed", None) is not None and m._w16_version == ver and m._w16_device == device:
        return m._w16_packed
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    hid = int(C_ * m.mlp_ratio)
    HP = _rup(hid, 64)
    P: Dict[str, torch.Tensor] = {}
    with torch.no_grad(), batched_pack() as pk:
        for li, layer in enumerate(m.layers):
            nH = m.heads[li]
            dh, CA = C_ // nH, nH * 32
            hm = _head_map(nH, dh, device)
            qkv_rows = _qkv_rows(nH, dh, device)
            for bi, blk in enumerate(layer.residual_group.blocks):
                pre = f"{li}.{bi}."
                P[pre + "Wqkv"] = _pack_linear(blk.attn.qkv.weight, 3 * CA, CP, row_map=qkv_rows)
                P[pre + "bqkv"] = _pack_vec(blk.attn.qkv.bias, 3 * CA, row_map=qkv_rows)
                P[pre + "Wproj"] = _pack_linear(blk.attn.proj.weight, CP, CA, col_map=hm)
                P[pre + "bproj"] = _pack_vec(blk.attn.proj.bias, CP)
                P[pre + "W1"] = _pack_linear(blk.mlp.fc1.weight, HP, CP)
                P[pre + "b1"] = _pack_vec(blk.mlp.fc1.bias, HP)
                P[pre + "W2"] = _pack_linear(blk.mlp.fc2.weight, CP, HP)
                P[pre + "b2"] = _pack_vec(blk.mlp.fc2.bias, CP)
            P[f"{li}.Wconv"] = _pack_conv(layer.conv.weight, CP, CP)
            P[f"{li}.bconv"] = _pack_vec(layer.conv.bias, CP)
        P["Wcab"] = _pack_conv(m.conv_after_body.weight, CP, CP)
        P["bcab"] = _pack_vec(m.conv_after_body.bias, CP)
        if m.upsampler == "pixelshuffle":
            P["Wbefore"] = _pack_conv(m.conv_before_upsample[0].weight, 64, CP)
            P["bbefore"] = _pack_vec(m.conv_before_upsample[0].bias, 64)
            k = 0
            for mod in m.upsample:
                if isinstance(mod, nn.Conv2d):
                    r = int(round(math.sqrt(mod.weight.shape[0] // 64)))
                    pm = _ps_map(mod.weight.shape[0], r, 64, device)
                    P[f"Wup{k}"] = _pack_conv(mod.weight, mod.weight.shape[0], 64, row_map=pm)
                    P[f"bup{k}"] = _pack_vec(mod.bias, mod.weight.shape[0], row_map=pm)
                    P[f"rup{k}"] = torch.tensor(r)
                    k += 1
            P["Wlast"] = _pack_conv(m.conv_last.weight, 16, 64)
            P["blast"] = _pack_vec(m.conv_last.bias, 16)
        else:
            P["Wdirect"] = _pack_conv(m.upsample[0].weight, 16, CP)
            P["bdirect"] = _pack_vec(m.upsample[0].bias, 16)
        pk.resolve(P)
    m._w16_packed, m._w16_version, m._w16_device = P, ver, device
    return P


def forward(m, x: torch.Tensor) -> torch.Tensor:
    why = unsupported_reason(m)
    if why:
        raise SrkUnsupported(f"the MI355X HIP path does not cover {why}; no fallback path exists in this package")
    p0 = next(m.parameters())
    if p0.device != x.device:
        raise RuntimeError(f"input is on {x.device} but the model is on {p0.device}")
    _lib.claim_device(x.device.index if x.device.index is not None else torch.cuda.current_device())
    with torch.no_grad(), torch.cuda.device(x.device):
        return _forward(m, x.contiguous().float(), pack(m, x.device))


def _forward(m, x: torch.Tensor, P: Dict[str, torch.Tensor]) -> torch.Tensor:
    dev = x.device
    st = torch.cuda.current_stream(dev).cuda_stream
    B, Cin, H0, W0 = x.shape
    ws, s = 16, m.upscale
    H, W = _rup(H0, ws), _rup(W0, ws)
    if (H - H0 >= H0) or (W - W0 >= W0):
        raise RuntimeError(f"reflect padding {H0}x{W0} -> {H}x{W} needs pad < size (as torch 'reflect')")
    T, HW = B * H * W, H * W
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    HP = _rup(int(C_ * m.mlp_ratio), 64)
    f32, b16 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
    L = lib()
    mean3 = (C.c_float * 3)(*(m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]))
    img4 = torch.empty(T, 4, **f32)
    check(L.srk_img_prep(x.data_ptr(), img4.data_ptr(), B, Cin, H0, W0, H, W, float(m.img_range), C.byref(mean3), st))
    f0 = torch.empty(T, CP, **f32)
    check(L.srk_stem_conv(img4.data_ptr(), m.conv_first.weight.data_ptr(), m.conv_first.bias.data_ptr(), f0.data_ptr(), B, H, W, Cin, C_, CP, st))
    _, cur, _, _ = ops.layernorm_fwd(f0, m.patch_embed.norm.weight, m.patch_embed.norm.bias, C_, out_bf16=False, out_f32=True)

    CAmax = max(h * 32 for h in m.heads)
    qkv, ao = torch.empty(T, 3 * CAmax, **b16), torch.empty(T, CAmax, **b16)
    xn2, hh, xb = torch.empty(T, CP, **b16), torch.empty(T, HP, **b16), torch.empty(T, CP, **b16)
    stat_a, stat_b = torch.empty(T, **f32), torch.empty(T, **f32)
    xn_a, xn_b = torch.empty(T, CP, **b16), torch.empty(T, CP, **b16)
    fused_mlp_ok = (CP == 192 and HP == 384 and T % 64 == 0 and T >= 64 * torch.cuda.get_device_properties(dev).multi_processor_count)
    ln_fusable = CP in (64, 128, 192)

    def next_norm(norm, dst):
        return dict(out=dst, mean=stat_a, rstd=stat_b, gamma=norm.weight, beta=norm.bias, C=C_) if ln_fusable else None

    def mlp(pre, xn_in, x_res, out, out_b=None, nn_=None):
        if fused_mlp_ok:
            args = (None, None, None, None, None, 0) if nn_ is None else (nn_["out"].data_ptr(), nn_["mean"].data_ptr(), nn_["rstd"].data_ptr(),
                                                                       nn_["gamma"].data_ptr(), nn_["beta"].data_ptr(), nn_["C"])
            check(L.srk_mlp_fused_fwd(xn_in.data_ptr(), P[pre + "W1"].data_ptr(), P[pre + "b1"].data_ptr(), P[pre + "W2"].data_ptr(),
                                      P[pre + "b2"].data_ptr(), x_res.data_ptr(), out.data_ptr(), _ptr(out_b), *args, T, st))
        else:
            _gemm(st, _lib.LD_ROWS, _lib.EP_GELU, xn_in, P[pre + "W1"], T, HP, CP, lda=CP, bias=P[pre + "b1"], outb2=hh)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, hh, P[pre + "W2"], T, CP, HP, lda=HP, bias=P[pre + "b2"], res=x_res, outf=out, outb=out_b, xn=nn_)

    xn1 = None
    for li, layer in enumerate(m.layers):
        nH = m.heads[li]
        CA = nH * 32
        scale = float(m.qk_scale or (C_ // nH) ** -0.5)
        layer_in = cur
        blocks = list(layer.residual_group.blocks)
        for bi, blk in enumerate(blocks):
            pre = f"{li}.{bi}."
            if xn1 is None:
                xn1, _, _, _ = ops.layernorm_fwd(cur, blk.norm1.weight, blk.norm1.bias, C_)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
            tab = blk.attn.relative_position_bias_table
            sh = blk.shift_size
            check(L.srk_win256_attention_fwd(qkv.data_ptr(), 3 * CA, CA, tab.data_ptr(), tab.shape[0], ao.data_ptr(), CA, B, H, W, ws, ws, sh, sh, nH,
                                             scale, 0, st))
            x1 = torch.empty(T, CP, **f32)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, ao, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1,
                  xn=dict(out=xn2, mean=stat_a, rstd=stat_b, gamma=blk.norm2.weight, beta=blk.norm2.bias, C=C_))
            nxt = torch.empty(T, CP, **f32)
            last = bi == len(blocks) - 1
            dst = xn_a if xn1 is not xn_a else xn_b
            nn_ = None if last else next_norm(blocks[bi + 1].norm1, dst)
            mlp(pre, xn2, x1, nxt, out_b=xb if last else None, nn_=nn_)
            cur = nxt
            xn1 = dst if nn_ is not None else None
        nxt = torch.empty(T, CP, **f32)
        following = m.layers[li + 1].residual_group.blocks[0].norm1 if li + 1 < len(m.layers) else m.norm
        nn_ = next_norm(following, xn_a)
        _gemm(st, _lib.LD_CONV3, _lib.EP_RES, xb, P[f"{li}.Wconv"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P[f"{li}.bconv"], res=layer_in, outf=nxt,
              xn=nn_)
        cur = nxt
        xn1 = xn_a if nn_ is not None else None

    xnf = xn1 if xn1 is not None else ops.layernorm_fwd(cur, m.norm.weight, m.norm.bias, C_)[0]
    fb = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_RES_BF16, xnf, P["Wcab"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P["bcab"], res=f0, outb=fb)
    y = torch.empty(B, Cin, H0 * s, W0 * s, **f32)
    mean4 = (m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]) + [0.0]
    img = dict(inv_range=1.0 / float(m.img_range), Cimg=Cin, Hc=H0 * s, Wc=W0 * s, mean=mean4)
    if m.upsampler == "pixelshuffle":
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
        _gemm(st, _lib.LD_CONV3, _lib.EP_PS_IMG, fb, P["Wdirect"], T, 16, 9 * CP, conv=(B, H, W, CP), bias=P["bdirect"], outf=y, img=dict(img), r=s)
    return y
