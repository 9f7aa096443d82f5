"""HAT training on MI355X: the forward that keeps what the backward needs, and the backward pass, both as host-side sequences of
C-ABI calls (include/srk.h) -- the training-mode counterpart of ``hat_arch._hat_forward``.

Reference: hat_arch.py:281-325 (HAB.forward), :403-439 (OCAB.forward), :600-620 (RHAG), :943-987 (HAT.forward); autograd of those
is what ``hat_backward`` restates by hand, block by block in reverse:

    tail      conv_last (fp32 small-conv gradients), conv + PixelShuffle stages (weight gradient on the shuffled gradient, dgrad
              through the pixel-shuffled loader), conv_before_upsample (+ LeakyReLU'), conv_after_body, final LayerNorm
    RHAG      conv dgrad / wgrad, RSTB-style skip add at the layer input
    OCAB/HAB  fc2 dgrad * GELU'(u) -> fc1 dgrad -> LayerNorm backward (adds into the fp32 gradient stream) ; CAB: channel gate,
              both 3x3 convs ; proj dgrad -> 256-query window attention backward (csrc/attn256_bwd.hip) -> qkv dgrad (+ the conv
              branch's gradient) -> LayerNorm backward
    head      patch_embed.norm backward + long skip, conv_first weight gradient

DropPath (drop_path_rate > 0 in train mode, hat_arch.py:258 / :321-325) is data: per-block, per-sample factors 0 or 1 / keep
that scale the attention and MLP branch in the forward epilogues and the bf16 gradient copies that enter those branches.
The weight gradients come out of the kernels in the packed (padded / permuted) layouts and are scattered back into the
parameters' shapes with index maps (host plumbing on parameter-sized tensors).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

import os

from . import _lib, ops
from ._lib import check, lib


_POISON = os.environ.get("SRK_DBG_POISON") == "1"


def _ha():
    from . import hat_arch
    return hat_arch


# ---- packed operands of the backward pass (transposed copies for the dgrads) ---------------------------------------------------
def pack_transposed(m, device) -> Dict[str, torch.Tensor]:
    ha = _ha()
    ver = sum(p._version for p in m.parameters())
    if getattr(m, "_packedT", None) is not None and m._packedT_version == ver and m._packedT_device == device:
        return m._packedT
    C_, CP = m.embed_dim, ha._rup(m.embed_dim, 64)
    HP = ha._rup(int(C_ * m.mlp_ratio), 64)
    P: Dict[str, torch.Tensor] = {}
    _pack_conv_T = ha._pack_conv_T
    with torch.no_grad(), ha.batched_pack() as pk:
        for li, layer in enumerate(m.layers):
            nH = m.heads[li]
            dh, CA = C_ // nH, nH * 32
            hm = ha._head_map(nH, dh, device)
            qkv_rows = ha._qkv_rows(nH, dh, device)

            def attn_T(pre, qkv, proj):
                P[pre + "WqkvT"] = ha._pack_linear(qkv.weight.t(), CP, 3 * CA, col_map=qkv_rows)        # [c][3 CA]
                P[pre + "WprojT"] = ha._pack_linear(proj.weight.t(), CA, CP, row_map=hm)                # [ca][c]

            def mlp_T(pre, mlp):
                P[pre + "W1T"] = ha._pack_linear(mlp.fc1.weight.t(), CP, HP)
                P[pre + "W2T"] = ha._pack_linear(mlp.fc2.weight.t(), HP, CP)

            for bi, blk in enumerate(layer.residual_group.blocks):
                pre = f"{li}.{bi}."
                attn_T(pre, blk.attn.qkv, blk.attn.proj)
                mlp_T(pre, blk.mlp)
                cab = blk.conv_block.cab
                P[pre + "Wc0T"] = _pack_conv_T(cab[0].weight, CP, 64)
                P[pre + "Wc2T"] = _pack_conv_T(cab[2].weight, 64, CP)
            oc = layer.residual_group.overlap_attn
            attn_T(f"{li}.oca.", oc.qkv, oc.proj)
            mlp_T(f"{li}.oca.", oc.mlp)
            P[f"{li}.WconvT"] = _pack_conv_T(layer.conv.weight, CP, CP)
        P["WcabT"] = _pack_conv_T(m.conv_after_body.weight, CP, CP)
        P["WbeforeT"] = _pack_conv_T(m.conv_before_upsample[0].weight, CP, 64)
        k = 0
        for mod in m.upsample:
            if isinstance(mod, nn.Conv2d):
                r = int(round(math.sqrt(mod.weight.shape[0] // 64)))
                pm = ha._ps_map(mod.weight.shape[0], r, 64, device)
                P[f"WupT{k}"] = _pack_conv_T(mod.weight, 64, mod.weight.shape[0], col_map=pm)
                k += 1
        pk.resolve(P)
    m._packedT, m._packedT_version, m._packedT_device = P, ver, device
    return P


# ---- forward, keeping activations ---------------------------------------------------------------------------------------------
def hat_forward_train(m, x: torch.Tensor, P: Dict[str, torch.Tensor], drop: Optional[torch.Tensor]) -> (torch.Tensor, dict):
    """drop: None or fp32 [n_blocks][B] DropPath factors (0 or 1 / keep_prob) shared by a HAB's attention and MLP branch
    (hat_arch.py:321-325 draws them independently; here each branch gets its own row: [n_blocks][2][B])."""
    ha = _ha()
    _gemm, _rup, _ptr = ha._gemm, ha._rup, ha._ptr
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
    S: dict = dict(B=B, Cin=Cin, H0=H0, W0=W0, H=H, W=W, T=T, blocks=[], layers=[], drop=drop)

    mean3 = (C.c_float * 3)(*(m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]))
    img4 = torch.empty(T, 4, **f32)
    check(L.srk_img_prep(x.data_ptr(), img4.data_ptr(), B, Cin, H0, W0, H, W, float(m.img_range), C.byref(mean3), st))
    f0 = torch.empty(T, CP, **f32)
    check(L.srk_stem_conv(img4.data_ptr(), m.conv_first.weight.data_ptr(), m.conv_first.bias.data_ptr(), f0.data_ptr(), B, H, W, Cin, C_, CP, st))
    _, cur, mean_pe, rstd_pe = ops.layernorm_fwd(f0, m.patch_embed.norm.weight, m.patch_embed.norm.bias, C_, out_bf16=False, out_f32=True)
    S.update(img4=img4, f0=f0, mean_pe=mean_pe, rstd_pe=rstd_pe)

    gate_ws = torch.empty(max(1, int(L.srk_channel_gate_workspace(B, HW, CP))), dtype=torch.uint8, device=dev)
    fused_mlp_ok = (CP == 192 and HP == 384 and T % 64 == 0 and T >= 64 * torch.cuda.get_device_properties(dev).multi_processor_count)

    def rs(bidx, which):
        return None if drop is None else drop[bidx, which]

    def mlp(pre, xn_in, x_res, rowscale):
        """-> (out fp32, out bf16, u, h): out = x_res + f * fc2(gelu(fc1(xn_in)))"""
        out = torch.empty(T, CP, **f32)
        out_b = torch.empty(T, CP, **b16)
        u, h = torch.empty(T, HP, **b16), torch.empty(T, HP, **b16)
        if fused_mlp_ok:
            check(L.srk_mlp_fused_fwd_train(xn_in.data_ptr(), P[pre + "W1"].data_ptr(), P[pre + "b1"].data_ptr(), P[pre + "W2"].data_ptr(),
                                            P[pre + "b2"].data_ptr(), x_res.data_ptr(), out.data_ptr(), out_b.data_ptr(), u.data_ptr(),
                                            h.data_ptr(), None, None, None, None, None, 0, _ptr(rowscale), HW, T, st))
        else:
            _gemm(st, _lib.LD_ROWS, _lib.EP_GELU, xn_in, P[pre + "W1"], T, HP, CP, lda=CP, bias=P[pre + "b1"], outb=u, outb2=h)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, h, P[pre + "W2"], T, CP, HP, lda=HP, bias=P[pre + "b2"], res=x_res, outf=out, outb=out_b,
                  rowscale=rowscale, rows_per_sample=HW)
        return out, out_b, u, h

    bidx = 0
    for li, layer in enumerate(m.layers):
        nH = m.heads[li]
        CA = nH * 32
        scale = float(m.qk_scale or (C_ // nH) ** -0.5)
        layer_in = cur
        oc = layer.residual_group.overlap_attn
        for bi, blk in enumerate(layer.residual_group.blocks):
            pre = f"{li}.{bi}."
            xn1, _, mean1, rstd1 = ops.layernorm_fwd(cur, blk.norm1.weight, blk.norm1.bias, C_)
            qkv = torch.empty(T, 3 * CA, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
            tab = blk.attn.relative_position_bias_table
            sh = blk.shift_size
            ao = torch.empty(T, CA, **b16)
            check(L.srk_win256_attention_fwd(qkv.data_ptr(), 3 * CA, CA, tab.data_ptr(), tab.shape[0], ao.data_ptr(), CA, B, H, W, ws, ws,
                                             sh, sh, nH, scale, 0, st))
            x1 = torch.empty(T, CP, **f32)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, ao, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1,
                  rowscale=rs(bidx, 0), rows_per_sample=HW)
            u1, c1 = torch.empty(T, 64, **b16), torch.empty(T, 64, **b16)
            _gemm(st, _lib.LD_CONV3, _lib.EP_GELU, xn1, P[pre + "Wc0"], T, 64, 9 * CP, conv=(B, H, W, CP), bias=P[pre + "bc0"], outb=u1, outb2=c1)
            c2 = torch.empty(T, CP, **b16)
            _gemm(st, _lib.LD_CONV3, _lib.EP_BF16, c1, P[pre + "Wc2"], T, CP, 9 * 64, conv=(B, H, W, 64), bias=P[pre + "bc2"], outb=c2)
            gate = torch.empty(B, CP, **f32)
            Sq = P[pre + "ca_w1"].shape[0]
            check(L.srk_channel_gate(c2.data_ptr(), gate_ws.data_ptr(), P[pre + "ca_w1"].data_ptr(), P[pre + "ca_b1"].data_ptr(),
                                     P[pre + "ca_w2"].data_ptr(), P[pre + "ca_b2"].data_ptr(), float(blk.conv_scale), gate.data_ptr(), B, HW,
                                     C_, CP, Sq, st))
            check(L.srk_cab_add_ln(x1.data_ptr(), c2.data_ptr(), gate.data_ptr(), None, None, None, T, HW, C_, CP, st))     # x1 += conv * gate
            xn2, _, mean2, rstd2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, C_)
            nxt, _, u, h = mlp(pre, xn2, x1, rs(bidx, 1))
            S["blocks"].append(dict(kind="hab", li=li, bi=bi, pre=pre, blk=blk, nH=nH, CA=CA, scale=scale, shift=sh, x_in=cur, xn1=xn1,
                                    mean1=mean1, rstd1=rstd1, qkv=qkv, ao=ao, x1=x1, u1=u1, c1=c1, c2=c2, gate=gate, xn2=xn2, mean2=mean2,
                                    rstd2=rstd2, u=u, h=h, bidx=bidx))
            cur = nxt
            bidx += 1
        pre = f"{li}.oca."
        xn1, _, mean1, rstd1 = ops.layernorm_fwd(cur, oc.norm1.weight, oc.norm1.bias, C_)
        qkv = torch.empty(T, 3 * CA, **b16)
        _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
        tab = oc.relative_position_bias_table
        ao = torch.empty(T, CA, **b16)
        check(L.srk_win256_attention_fwd(qkv.data_ptr(), 3 * CA, CA, tab.data_ptr(), tab.shape[0], ao.data_ptr(), CA, B, H, W, ws, ws, 0, 0, nH,
                                         scale, oc.overlap_win_size - ws, st))
        x1 = torch.empty(T, CP, **f32)
        _gemm(st, _lib.LD_ROWS, _lib.EP_RES, ao, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1)
        xn2, _, mean2, rstd2 = ops.layernorm_fwd(x1, oc.norm2.weight, oc.norm2.bias, C_)
        x2, xb, u, h = mlp(pre, xn2, x1, None)
        S["blocks"].append(dict(kind="ocab", li=li, pre=pre, blk=oc, nH=nH, CA=CA, scale=scale, x_in=cur, xn1=xn1, mean1=mean1, rstd1=rstd1,
                                qkv=qkv, ao=ao, x1=x1, xn2=xn2, mean2=mean2, rstd2=rstd2, u=u, h=h))
        nxt = torch.empty(T, CP, **f32)
        _gemm(st, _lib.LD_CONV3, _lib.EP_RES, xb, P[f"{li}.Wconv"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P[f"{li}.bconv"], res=layer_in, outf=nxt)
        S["layers"].append(dict(li=li, xb=xb, n_blocks=len(layer.residual_group.blocks) + 1))
        cur = nxt

    xnf, _, meanf, rstdf = ops.layernorm_fwd(cur, m.norm.weight, m.norm.bias, C_)
    fb = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_RES_BF16, xnf, P["Wcab"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P["bcab"], res=f0, outb=fb)
    t1 = torch.empty(T, 64, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_LRELU, fb, P["Wbefore"], T, 64, 9 * CP, conv=(B, H, W, CP), bias=P["bbefore"], outb=t1, scale=0.01)
    S.update(x_last=cur, xnf=xnf, meanf=meanf, rstdf=rstdf, fb=fb, t1=t1, ups=[])
    src, h_, w_ = t1, H, W
    k = 0
    while f"Wup{k}" in P:
        r = int(P[f"rup{k}"])
        N = P[f"Wup{k}"].shape[0]
        up = torch.empty(B * h_ * r * w_ * r, 64, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_PS, src, P[f"Wup{k}"], B * h_ * w_, N, 9 * 64, conv=(B, h_, w_, 64), bias=P[f"bup{k}"], outb=up, r=r, Cs=64,
              ldo=N)
        S["ups"].append(dict(src=src, out=up, h=h_, w=w_, r=r, N=N))
        src, h_, w_, k = up, h_ * r, w_ * r, k + 1
    y = torch.empty(B, Cin, H0 * s, W0 * s, **f32)
    mean4 = (m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]) + [0.0]
    _gemm(st, _lib.LD_CONV3, _lib.EP_IMG, src, P["Wlast"], B * h_ * w_, 16, 9 * 64, conv=(B, h_, w_, 64), bias=P["blast"], outf=y,
          img=dict(inv_range=1.0 / float(m.img_range), Cimg=Cin, Hc=H0 * s, Wc=W0 * s, mean=mean4))
    S.update(hr_h=h_, hr_w=w_)
    return y, S


# ---- backward -----------------------------------------------------------------------------------------------------------------------
_ARANGE: Dict[tuple, torch.Tensor] = {}


def _arange(n: int, device) -> torch.Tensor:
    key = (n, str(device))
    t = _ARANGE.get(key)
    if t is None:
        t = _ARANGE[key] = torch.arange(n, device=device)
    return t


def _unpack_linear(dw: torch.Tensor, N: int, K: int, row_map=None, col_map=None) -> torch.Tensor:
    if row_map is None and col_map is None:
        return dw[:N, :K].contiguous()
    rows = row_map if row_map is not None else _arange(N, dw.device)
    cols = col_map if col_map is not None else _arange(K, dw.device)
    return dw[rows[:, None], cols[None, :]].contiguous()


def _unpack_conv(dw: torch.Tensor, Cout: int, Cin: int, CinP: int, row_map=None) -> torch.Tensor:
    v = dw.view(dw.shape[0], 9, CinP)
    v = v[:Cout] if row_map is None else v[row_map]
    return v[:, :, :Cin].reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()


def hat_backward(m, S: dict, dy: torch.Tensor, hook=None) -> Dict[str, torch.Tensor]:
    """-> {parameter name: gradient} for every parameter of the model.  hook (distributed.ListGradSynchronizer or None): gets the
    gradient tensors of each finished segment (tail, every RHAG, head) so that their all-reduce overlaps the next segment."""
    ha = _ha()
    _gemm, _rup, _ptr = ha._gemm, ha._rup, ha._ptr
    P = m._pack(dy.device)
    PT = pack_transposed(m, dy.device)
    dev = dy.device
    st = torch.cuda.current_stream(dev).cuda_stream
    B, Cin, H0, W0, H, W, T = S["B"], S["Cin"], S["H0"], S["W0"], S["H"], S["W"], S["T"]
    HW = H * W
    s = m.upscale
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    hid = int(C_ * m.mlp_ratio)
    HP = _rup(hid, 64)
    f32 = dict(dtype=torch.float32, device=dev)
    b16 = dict(dtype=torch.bfloat16, device=dev)
    L = lib()
    drop = S["drop"]
    G: Dict[str, torch.Tensor] = {}
    names = {id(p): n for n, p in m.named_parameters()}
    handed = set()

    def segment_done():
        if hook is not None:
            fresh = [k for k in G if k not in handed]
            handed.update(fresh)
            hook.segment_done([G[k] for k in fresh])

    def pname(p):
        return names[id(p)]

    pending = []          # the block's linear weight gradients: queued, then ONE launch for all four (flush_wgrads)

    def lin_wgrad(y, x, lin, NP, KP, row_map=None, col_map=None, prefix=None):
        """dW += y^T x, db += colsum(y) in the packed layout -> the nn.Linear's gradient (computed at the next flush_wgrads: y and x must
        stay untouched until then)."""
        pending.append((y, x, lin, row_map, col_map))

    def flush_wgrads():
        if not pending:
            return
        for (y, x, lin, row_map, col_map), (dw, db) in zip(pending, ops.linear_wgrad_multi_bf16([(q[0], q[1]) for q in pending])):
            N, K = lin.weight.shape
            G[pname(lin.weight)] = _unpack_linear(dw, N, K, row_map, col_map)
            if lin.bias is not None:
                G[pname(lin.bias)] = (db[:N] if row_map is None else db[row_map]).contiguous()
        pending.clear()

    def conv_wgrad(dyb, xb, conv, Bc, Hc, Wc, CinP, NP, r=1, row_map=None):
        dw = ops.zeros_f32((NP, 9 * CinP), dev)
        db = ops.zeros_f32((NP,), dev)
        ops._bind_wgrad_workspace(dev)
        if r == 1:
            check(L.srk_conv3x3_wgrad_bf16(dyb.data_ptr(), xb.data_ptr(), dw.data_ptr(), db.data_ptr(), Bc, Hc, Wc, CinP, NP, st))
        else:
            check(L.srk_conv3x3_wgrad_ps_bf16(dyb.data_ptr(), xb.data_ptr(), dw.data_ptr(), db.data_ptr(), Bc, Hc, Wc, CinP, NP, r, 64, st))
        Cout, Cin_ = conv.weight.shape[:2]
        G[pname(conv.weight)] = _unpack_conv(dw, Cout, Cin_, CinP, row_map)
        G[pname(conv.bias)] = (db[:Cout] if row_map is None else db[row_map]).contiguous()

    def ln_bwd(dyb, x, mean, rstd, norm, gx, gxb, accumulate):
        dg, dbt = ops.zeros_f32((C_,), dev), ops.zeros_f32((C_,), dev)
        check(L.srk_layernorm_bwd(dyb.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), norm.weight.data_ptr(), gx.data_ptr(),
                                  _ptr(gxb), dg.data_ptr(), dbt.data_ptr(), T, C_, CP, 1 if accumulate else 0, st))
        G[pname(norm.weight)], G[pname(norm.bias)] = dg, dbt

    def scaled(gb, bidx, which):
        """bf16 gradient copy entering a branch whose output was scaled by a DropPath factor"""
        if drop is None:
            return gb
        out = torch.empty_like(gb)
        check(L.srk_rowscale_bf16(gb.data_ptr(), out.data_ptr(), drop[bidx, which].data_ptr(), T, HW, CP, st))
        return out

    # ---------------- reconstruction tail ----------------
    hs, wsz = S["hr_h"], S["hr_w"]
    gyimg = torch.empty(B * hs * wsz, 4, **f32)
    check(L.srk_img_grad_prep(dy.data_ptr(), gyimg.data_ptr(), B, Cin, H0 * s, W0 * s, hs, wsz, 1, 4, 1.0 / float(m.img_range), st))
    last_in = S["ups"][-1]["out"] if S["ups"] else S["t1"]
    dwl, dbl = torch.zeros_like(m.conv_last.weight, dtype=torch.float32), torch.zeros_like(m.conv_last.bias, dtype=torch.float32)
    check(L.srk_smallconv_wgrad(last_in.data_ptr(), gyimg.data_ptr(), dwl.data_ptr(), dbl.data_ptr(), B, hs, wsz, 64, 64, Cin, 4, st))
    G[pname(m.conv_last.weight)], G[pname(m.conv_last.bias)] = dwl, dbl
    gcur = torch.empty(B * hs * wsz, 64, **b16)
    check(L.srk_smallconv_dgrad(gyimg.data_ptr(), m.conv_last.weight.data_ptr(), gcur.data_ptr(), B, hs, wsz, 64, 64, Cin, 4, st))
    up_convs = [mod for mod in m.upsample if isinstance(mod, nn.Conv2d)]
    for k in range(len(S["ups"]) - 1, -1, -1):
        u = S["ups"][k]
        r, N, h_, w_ = u["r"], u["N"], u["h"], u["w"]
        pm = ha._ps_map(N, r, 64, dev)
        conv_wgrad(gcur, u["src"], up_convs[k], B, h_, w_, 64, N, r=r, row_map=pm)
        gprev = torch.empty(B * h_ * w_, 64, **b16)
        if k == 0:     # through the LeakyReLU(0.01) of conv_before_upsample
            _gemm(st, _lib.LD_CONV3_PS, _lib.EP_DLRELU, gcur, PT[f"WupT{k}"], B * h_ * w_, 64, 9 * N, conv=(B, h_, w_, N), r=r, Cs=64, outb=gprev,
                  aux=S["t1"], scale=0.01, ldo=64)
        else:
            _gemm(st, _lib.LD_CONV3_PS, _lib.EP_BF16, gcur, PT[f"WupT{k}"], B * h_ * w_, 64, 9 * N, conv=(B, h_, w_, N), r=r, Cs=64, outb=gprev, ldo=64)
        gcur = gprev
    gt1 = gcur
    conv_wgrad(gt1, S["fb"], m.conv_before_upsample[0], B, H, W, CP, 64)
    gfb = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_BF16, gt1, PT["WbeforeT"], T, CP, 9 * 64, conv=(B, H, W, 64), outb=gfb)
    conv_wgrad(gfb, S["xnf"], m.conv_after_body, B, H, W, CP, CP)
    dxn = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_BF16, gfb, PT["WcabT"], T, CP, 9 * CP, conv=(B, H, W, CP), outb=dxn)
    gx = torch.empty(T, CP, **f32)        # gradient of the current layer's OUTPUT (later: of its input)
    gxb = torch.empty(T, CP, **b16)
    ln_bwd(dxn, S["x_last"], S["meanf"], S["rstdf"], m.norm, gx, gxb, accumulate=False)
    segment_done()

    opt = C.c_int()
    check(L.srk_get_option(b"mlp_bwd_fused", C.byref(opt)))
    fused_mlp_bwd_ok = (opt.value != 0 and CP == 192 and HP == 384 and T % 64 == 0 and HW % 64 == 0 and
                        T >= 64 * torch.cuda.get_device_properties(dev).multi_processor_count)
    # ---------------- layers, last to first ----------------
    blocks = S["blocks"]
    pos = len(blocks)
    attn_scratch = None
    for lay in reversed(S["layers"]):
        li = lay["li"]
        layer = m.layers[li]
        conv_wgrad(gxb, lay["xb"], layer.conv, B, H, W, CP, CP)
        gx2 = torch.empty(T, CP, **f32)       # gradient stream through the layer's body
        gxb2 = torch.empty(T, CP, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_F32_BF16, gxb, PT[f"{li}.WconvT"], T, CP, 9 * CP, conv=(B, H, W, CP), outf=gx2, outb=gxb2)
        for _ in range(lay["n_blocks"]):
            pos -= 1
            bk = blocks[pos]
            pre, blk, nH, CA = bk["pre"], bk["blk"], bk["nH"], bk["CA"]
            hab = bk["kind"] == "hab"
            hm = ha._head_map(nH, C_ // nH, dev)
            qkv_rows = ha._qkv_rows(nH, C_ // nH, dev)
            # ---- MLP half: x2 = x1 + f_mlp * fc2(gelu(fc1(norm2(x1)))) ----
            g_mlp = scaled(gxb2, bk["bidx"], 1) if hab else gxb2
            du = torch.empty(T, HP, **b16)
            g1b = torch.empty(T, CP, **b16)
            g1b_scaled = False
            if fused_mlp_bwd_ok:
                # fc2 dgrad * GELU' -> fc1 dgrad -> norm2 backward in one kernel; the bf16 copy comes out already scaled by the attention
                # branch's DropPath factor
                dg, dbt = ops.zeros_f32((C_,), dev), ops.zeros_f32((C_,), dev)
                rsc = drop[bk["bidx"], 0] if (hab and drop is not None) else None
                check(L.srk_mlp_fused_bwd(g_mlp.data_ptr(), PT[pre + "W2T"].data_ptr(), bk["u"].data_ptr(), du.data_ptr(), PT[pre + "W1T"].data_ptr(),
                                          bk["x1"].data_ptr(), bk["mean2"].data_ptr(), bk["rstd2"].data_ptr(), blk.norm2.weight.data_ptr(),
                                          gx2.data_ptr(), g1b.data_ptr(), _ptr(rsc), HW, dg.data_ptr(), dbt.data_ptr(), C_, T, st))
                G[pname(blk.norm2.weight)], G[pname(blk.norm2.bias)] = dg, dbt
                g1b_scaled = True
            else:
                _gemm(st, _lib.LD_ROWS, _lib.EP_DGELU, g_mlp, PT[pre + "W2T"], T, HP, CP, lda=CP, aux=bk["u"], outb=du, ldo=HP)
            lin_wgrad(g_mlp, bk["h"], blk.mlp.fc2, CP, HP)
            lin_wgrad(du, bk["xn2"], blk.mlp.fc1, HP, CP)
            if not fused_mlp_bwd_ok:
                dxn2 = torch.empty(T, CP, **b16)
                _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, du, PT[pre + "W1T"], T, CP, HP, lda=HP, outb=dxn2)
                ln_bwd(dxn2, bk["x1"], bk["mean2"], bk["rstd2"], blk.norm2, gx2, g1b, accumulate=True)      # gx2 = d x1 (fp32), g1b its bf16 copy
            dxc = None
            if hab:
                # ---- CAB: x1 += conv2(gelu(conv1(xn1))) * gate ----
                cab = blk.conv_block.cab
                att = cab[3].attention
                Sq = att[1].weight.shape[0]
                dw1, db1 = ops.zeros_f32((Sq, C_), dev), ops.zeros_f32((Sq,), dev)
                dw2, db2 = ops.zeros_f32((C_, Sq), dev), ops.zeros_f32((C_,), dev)
                dmean = torch.empty(B, CP, **f32)
                dc2 = torch.empty(T, CP, **b16)
                wsb = torch.empty(max(1, int(L.srk_cab_bwd_workspace(B, HW, CP))), dtype=torch.uint8, device=dev)
                check(L.srk_cab_bwd(bk["c2"].data_ptr(), gx2.data_ptr(), bk["gate"].data_ptr(), wsb.data_ptr(), P[pre + "ca_w1"].data_ptr(),
                                    P[pre + "ca_b1"].data_ptr(), P[pre + "ca_w2"].data_ptr(), P[pre + "ca_b2"].data_ptr(), float(blk.conv_scale),
                                    dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), dmean.data_ptr(), dc2.data_ptr(), B, HW, C_,
                                    CP, Sq, st))
                G[pname(att[1].weight)], G[pname(att[1].bias)] = dw1.view_as(att[1].weight), db1
                G[pname(att[3].weight)], G[pname(att[3].bias)] = dw2.view_as(att[3].weight), db2
                conv_wgrad(dc2, bk["c1"], cab[2], B, H, W, 64, CP)
                du1 = torch.empty(T, 64, **b16)
                _gemm(st, _lib.LD_CONV3, _lib.EP_DGELU, dc2, PT[pre + "Wc2T"], T, 64, 9 * CP, conv=(B, H, W, CP), aux=bk["u1"], outb=du1, ldo=64)
                conv_wgrad(du1, bk["xn1"], cab[0], B, H, W, CP, 64)
                dxc = torch.empty(T, CP, **f32)
                _gemm(st, _lib.LD_CONV3, _lib.EP_F32_BF16, du1, PT[pre + "Wc0T"], T, CP, 9 * 64, conv=(B, H, W, 64), outf=dxc)
            # ---- attention half: x1 = x + f_attn * proj(attention(qkv(norm1(x)))) ----
            attn_mod = blk.attn if hab else blk
            g_att = scaled(g1b, bk["bidx"], 0) if (hab and not g1b_scaled) else g1b
            dao = torch.empty(T, CA, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, g_att, PT[pre + "WprojT"], T, CA, CP, lda=CP, outb=dao, ldo=CA)
            lin_wgrad(g_att, bk["ao"], attn_mod.proj, CP, CA, col_map=hm)
            tab = attn_mod.relative_position_bias_table
            overlap = 0 if hab else blk.overlap_win_size - m.window_size
            sh = bk["shift"] if hab else 0
            need = int(L.srk_win256_attention_bwd_scratch(B, H, W, nH, CA, tab.shape[0], overlap))
            if attn_scratch is None or attn_scratch.numel() < need:
                attn_scratch = torch.empty(need, dtype=torch.uint8, device=dev)
            dqkv = torch.empty(T, 3 * CA, **b16) if not _POISON else torch.full((T, 3 * CA), float("nan"), **b16)      # every element is written by the attention backward
            dtab = ops.zeros_f32(tab.shape, dev)
            check(L.srk_win256_attention_bwd(bk["qkv"].data_ptr(), 3 * CA, CA, tab.data_ptr(), tab.shape[0], dao.data_ptr(), CA, dqkv.data_ptr(),
                                             dtab.data_ptr(), attn_scratch.data_ptr(), B, H, W, sh, sh, nH, bk["scale"], overlap, st))
            G[pname(tab)] = dtab
            lin_wgrad(dqkv, bk["xn1"], attn_mod.qkv, 3 * CA, CP, row_map=qkv_rows)
            flush_wgrads()            # before the kernel below overwrites gxb2 (the fc2 gradient's operand when no DropPath copy was made)
            dxn1 = torch.empty(T, CP, **b16)
            if dxc is not None:
                _gemm(st, _lib.LD_ROWS, _lib.EP_RES_BF16, dqkv, PT[pre + "WqkvT"], T, CP, 3 * CA, lda=3 * CA, res=dxc, outb=dxn1)
                ln_bwd(dxn1, bk["x_in"], bk["mean1"], bk["rstd1"], blk.norm1, gx2, gxb2, accumulate=True)
            elif CP in (64, 128, 192):      # OCAB: qkv dgrad with the norm1 backward in its epilogue
                dg, dbt = ops.zeros_f32((C_,), dev), ops.zeros_f32((C_,), dev)
                _gemm(st, _lib.LD_ROWS, _lib.EP_LNBWD, dqkv, PT[pre + "WqkvT"], T, CP, 3 * CA, lda=3 * CA, outf=gx2, outb=gxb2, ldo=CP,
                      ln=dict(x=bk["x_in"], mean=bk["mean1"], rstd=bk["rstd1"], gamma=blk.norm1.weight, dgamma=dg, dbeta=dbt, C=C_))
                G[pname(blk.norm1.weight)], G[pname(blk.norm1.bias)] = dg, dbt
            else:
                _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, dqkv, PT[pre + "WqkvT"], T, CP, 3 * CA, lda=3 * CA, outb=dxn1)
                ln_bwd(dxn1, bk["x_in"], bk["mean1"], bk["rstd1"], blk.norm1, gx2, gxb2, accumulate=True)
        # layer skip: d(layer input) = d(body input) + d(layer output)
        check(L.srk_add_f32_bf16(gx.data_ptr(), gx2.data_ptr(), gxb.data_ptr(), T * CP, st))
        segment_done()

    # ---------------- head: patch_embed.norm, long skip, conv_first ----------------
    gf = torch.empty(T, CP, **f32)
    ln_bwd(gxb, S["f0"], S["mean_pe"], S["rstd_pe"], m.patch_embed.norm, gf, None, accumulate=False)
    check(L.srk_add_bf16_into_f32(gf.data_ptr(), gfb.data_ptr(), T * CP, st))
    dwf, dbf = torch.zeros_like(m.conv_first.weight, dtype=torch.float32), torch.zeros_like(m.conv_first.bias, dtype=torch.float32)
    check(L.srk_stem_wgrad(S["img4"].data_ptr(), gf.data_ptr(), dwf.data_ptr(), dbf.data_ptr(), B, H, W, Cin, C_, CP, st))
    G[pname(m.conv_first.weight)], G[pname(m.conv_first.bias)] = dwf, dbf
    segment_done()
    if hook is not None:
        hook.finish()
    return G


class HATFunction(torch.autograd.Function):
    """One autograd node for the whole model (as the SwinIR engine): forward keeps the activations, backward returns every
    parameter's gradient.  The input image gets no gradient."""

    @staticmethod
    def forward(ctx, model, x, drop, *params):
        with torch.cuda.device(x.device):
            y, saved = hat_forward_train(model, x.contiguous().float(), model._pack(x.device), drop)
        ctx.model, ctx.saved = model, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        model = ctx.model
        arena = model.__dict__.setdefault("_zero_arena", ops.ZeroArena())      # the pass's zeroed accumulators: one buffer, one fill
        with torch.cuda.device(dy.device), ops.arena_scope(arena, dy.device):
            G = hat_backward(model, ctx.saved, dy.contiguous().float(), hook=getattr(model, "grad_sync", None))
        ctx.saved = None
        grads = []
        for n, p in model.named_parameters():
            g = G.get(n)
            grads.append(None if g is None else g.reshape(p.shape).to(p.dtype))
        return (None, None, None, *grads)
