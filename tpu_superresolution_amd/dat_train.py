"""DAT training on MI355X: the forward that keeps what the backward needs (BatchNorm in training mode: batch statistics, running
statistics updated) and the backward pass, as host-side sequences of C-ABI calls (include/srk.h) -- the training-mode counterpart
of ``dat_arch._dat_forward``, in the manner of ``hat_train``.

Reference: dat_arch.py:366-446 (Adaptive_Spatial_Attention.forward), :481-528 (Adaptive_Channel_Attention.forward), :74-90 (SGFN),
:555-565 (DATB.forward: x + drop_path(attn(norm1 x)), x + drop_path(ffn(norm2 x))), :640-657 (ResidualGroup), :805-860 (DAT.forward).

Division of labour.  Everything that touches tokens runs in HIP kernels: the GEMMs / convs / LayerNorms of the SwinIR and HAT paths,
the rectangular-window attention (csrc/attn256.hip forward, csrc/attn_rect_bwd.hip backward) and csrc/dat_train.hip (token
reductions, per-channel affine maps, gating backward, depth-wise conv gradients, the channel attention's Gram products).  What sits
between two token passes is a function of a few hundred numbers -- BatchNorm's batch statistics -> scale / shift and their
backward coefficients (closed form below), channel_interaction on the pooled [B][C] vector, the d x d channel-attention softmax,
the DynamicPosBias MLP on the (2h-1)(2w-1) offsets -- and is evaluated here with torch on those small tensors (autograd for the
three small networks).  No token-sized tensor is ever touched by a torch op.

BatchNorm (training) for a channel with n values x: mu = sum x / n, var = sum x^2 / n - mu^2, rstd = (var + eps)^-1/2,
z = x * s + t with s = gamma * rstd, t = beta - mu * s.  Backward with S1 = sum dz, S2 = sum dz * x:
d beta = S1, d gamma = rstd * (S2 - mu * S1), d x = A dz + B x + C with A = s, B = -s * rstd * d gamma / n,
C = (s / n) * (mu * rstd * d gamma - S1).  The running estimates move by momentum 0.1 (unbiased variance), as nn.BatchNorm2d.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from . import _lib, ops
from ._lib import check, lib
from .hat_arch import _pack_conv_T
from .hat_train import _arange, _unpack_conv, _unpack_linear

EPS = 1e-5


def _da():
    from . import dat_arch
    return dat_arch


_POISON = os.environ.get("SRK_DBG_POISON") == "1"


def _ha():
    from . import hat_arch
    return hat_arch


# ---- small functions between token passes ------------------------------------------------------------------------------------------
def _bn_coeffs(s1: torch.Tensor, s2: torch.Tensor, n: int, gamma: torch.Tensor, beta: torch.Tensor):
    """sums of x and x^2 -> (scale, shift, mean, rstd, biased var)"""
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_min(0.0)
    rstd = torch.rsqrt(var + EPS)
    s = gamma * rstd
    return s, beta - mean * s, mean, rstd, var


def _bn_backward_coeffs(S1: torch.Tensor, S2: torch.Tensor, n: int, s: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor):
    """sums of dz and dz * x -> (A, B, C, d gamma, d beta) of d x = A dz + B x + C"""
    dgamma = rstd * (S2 - mean * S1)
    Bc = -s * rstd * dgamma / n
    Cc = (s / n) * (mean * rstd * dgamma - S1)
    return s, Bc, Cc, dgamma, S1


def _bn_forward(L, st, part: torch.Tensor, R: int, row_stride: int, ld: int, Cn: int, n: int, gamma: torch.Tensor, beta: torch.Tensor,
                bn: nn.BatchNorm2d, real_of: Optional[torch.Tensor]) -> torch.Tensor:
    """partial sums -> coef [4][ld] = scale, shift, mean, rstd in ONE launch (srk_bn_train_coeffs), running buffers moved in place"""
    coef = torch.empty(4, ld, dtype=torch.float32, device=part.device)
    track = bn.track_running_stats and bn.running_mean is not None and bn.momentum is not None and bn.running_mean.dtype == torch.float32
    check(L.srk_bn_train_coeffs(part.data_ptr(), R, row_stride, ld, Cn, float(n), gamma.data_ptr(), beta.data_ptr(), float(bn.eps), coef.data_ptr(),
                                bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                float(bn.momentum) if track else 0.0, None if real_of is None else real_of.data_ptr(), st))
    if track:
        bn.num_batches_tracked += 1
    elif bn.track_running_stats and bn.running_mean is not None:      # momentum None (cumulative average) / other dtypes: the torch path
        cm = coef[2, :Cn] if real_of is None else coef[2][real_of_inverse(real_of)]
        rs = coef[3, :Cn] if real_of is None else coef[3][real_of_inverse(real_of)]
        _bn_update(bn, cm, (1.0 / (rs * rs) - bn.eps).clamp_min(0.0), n)
    return coef


def real_of_inverse(real_of: torch.Tensor) -> torch.Tensor:
    """padded position of every real channel (the positions where real_of >= 0, ordered by the real index)"""
    pos = torch.nonzero(real_of >= 0).flatten()
    return pos[torch.argsort(real_of[pos])]


def _bn_update(bn: nn.BatchNorm2d, mean: torch.Tensor, var: torch.Tensor, n: int) -> None:
    """nn.BatchNorm2d's buffer update in training (momentum None = cumulative average, as torch)"""
    if not bn.track_running_stats or bn.running_mean is None:
        return
    bn.num_batches_tracked += 1
    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    bn.running_mean.mul_(1.0 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
    bn.running_var.mul_(1.0 - mom).add_((var * (n / max(n - 1, 1))).to(bn.running_var.dtype), alpha=mom)


def _channel_interaction(pm: torch.Tensor, ci: nn.Sequential, stats: Optional[dict] = None) -> torch.Tensor:
    """pooled mean [B][C] -> sigmoid(channel map) [B][C]  (dat_arch.py:315-321 on a 1 x 1 map, BatchNorm over the batch)"""
    y = F.linear(pm, ci[1].weight.flatten(1), ci[1].bias)
    n = y.shape[0]
    if n <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(y.shape) + (1, 1)}")
    mean = y.mean(0)
    var = y.var(0, unbiased=False)
    if stats is not None:
        stats["mean"], stats["var"], stats["n"] = mean.detach(), var.detach(), n
    y = (y - mean) * torch.rsqrt(var + ci[2].eps) * ci[2].weight + ci[2].bias
    return torch.sigmoid(F.linear(F.gelu(y), ci[4].weight.flatten(1), ci[4].bias))


def _sum_rows(part: torch.Tensor, outer: int, R: int) -> torch.Tensor:
    """contiguous fp32 [outer][R][...] -> [outer][...] (outer == 1: [...]), rows added in a fixed order by ONE small launch (srk_sum_rows_f32)"""
    assert part.is_contiguous() and part.dtype == torch.float32
    n = part.numel() // (outer * R)
    out = torch.empty(outer, n, dtype=torch.float32, device=part.device)
    check(lib().srk_sum_rows_f32(part.data_ptr(), outer, R, n, out.data_ptr(), torch.cuda.current_stream(part.device).cuda_stream))
    return out


def _ci_fused_ok(B: int, C: int, S: int, ci: nn.Sequential) -> bool:
    """srk_channel_interaction_fwd / _bwd cover this shape (csrc/dat_small.hip limits) and the BatchNorm has a fixed momentum"""
    bn = ci[2]
    return (bool(lib().srk_channel_interaction_covered(B, C, S)) and bn.affine and ci[1].weight.dtype == torch.float32
            and (not bn.track_running_stats or bn.running_mean is None or (bn.momentum is not None and bn.running_mean.dtype == torch.float32)))


def _hm32(hm: torch.Tensor) -> torch.Tensor:
    """int32 copy of a head map (padded position of every real channel), cached per map"""
    from . import hat_arch
    return hat_arch._cached_map(("hm32", hm.data_ptr(), hm.numel()), lambda: hm.to(torch.int32))


def _channel_attention_matrix(G: torch.Tensor, sq: torch.Tensor, sk: torch.Tensor, temp: torch.Tensor, dh: int) -> torch.Tensor:
    """G [B][h][32][32] = q^T k, sq / sk = column norms^2 of q / k -> softmax((q / |q|)^T (k / |k|) * temperature) over the dh real
    key channels, rows and columns of the padding zero (dat_arch.py:497-503; F.normalize clamps the norm at 1e-12)"""
    nq = sq[..., :dh].clamp_min(1e-24).sqrt()          # only the real channels enter the graph (the padding's zero norms would put 0 / 0 into
    nk = sk[..., :dh].clamp_min(1e-24).sqrt()          # the backward of the square root)
    logits = G[..., :dh, :dh] / (nq[..., :, None] * nk[..., None, :]) * temp.reshape(1, -1, 1, 1)
    return F.pad(torch.softmax(logits, dim=-1), (0, 32 - dh, 0, 32 - dh))


_POS_KEYS = ("pos_proj.weight", "pos_proj.bias", "pos1.0.weight", "pos1.0.bias", "pos1.2.weight", "pos1.2.bias", "pos2.0.weight", "pos2.0.bias",
             "pos2.2.weight", "pos2.2.bias", "pos3.0.weight", "pos3.0.bias", "pos3.2.weight", "pos3.2.bias")


def _pos_mlp_batched(x: torch.Tensor, Pm: Dict[str, torch.Tensor]) -> torch.Tensor:
    """DynamicPosBias (residual=False, dat_arch.py:93-130) of n position-bias modules at once: x [n][R][2], every parameter stacked along
    dim 0 -> [n][R][heads].  One batched op per layer instead of one per module and layer (36 modules in DAT x4)."""
    p = torch.baddbmm(Pm["pos_proj.bias"][:, None, :], x, Pm["pos_proj.weight"].transpose(1, 2))
    for k in ("pos1", "pos2", "pos3"):
        p = F.layer_norm(p, (p.shape[-1],)) * Pm[k + ".0.weight"][:, None, :] + Pm[k + ".0.bias"][:, None, :]
        p = torch.baddbmm(Pm[k + ".2.bias"][:, None, :], F.relu(p), Pm[k + ".2.weight"].transpose(1, 2))
    return p


class _PosBiasGroup:
    """All Spatial_Attention position-bias modules of a model with the same shapes: their dense biases [n][heads][N][N] from one
    batched evaluation, and -- after the backward has filled d_bias -- every module's parameter gradients from one batched autograd."""

    def __init__(self, sas):
        self.sas = sas
        self.index = {id(sa): i for i, sa in enumerate(sas)}
        self.rpe = torch.stack([sa.rpe_biases.float() for sa in sas])                               # [n][R][2]
        self.rpi = torch.stack([sa.relative_position_index.reshape(-1) for sa in sas])              # [n][N * N]
        self.N = sas[0].H_sp * sas[0].W_sp
        self.dense = self.d_bias = None

    def stacked(self):
        mods = [dict(sa.pos.named_parameters()) for sa in self.sas]
        return {k: torch.stack([mp[k].float() for mp in mods]) for k in _POS_KEYS}

    def forward(self):
        pos = _pos_mlp_batched(self.rpe, self.stacked())                                            # [n][R][hb]
        n, hb = pos.shape[0], pos.shape[2]
        g = pos.gather(1, self.rpi[:, :, None].expand(-1, -1, hb))                                   # [n][N * N][hb]
        self.dense = g.view(n, self.N, self.N, hb).permute(0, 3, 1, 2).contiguous()
        return self.dense

    def backward(self, put):
        n, hb = self.d_bias.shape[0], self.d_bias.shape[1]
        src = self.d_bias.permute(0, 2, 3, 1).reshape(n, self.N * self.N, hb)
        dpos = torch.zeros(n, self.rpe.shape[1], hb, dtype=torch.float32, device=src.device)
        dpos.scatter_add_(1, self.rpi[:, :, None].expand(-1, -1, hb), src)
        with torch.enable_grad():
            Pm = {k: v.detach().requires_grad_(True) for k, v in self.stacked().items()}
            grads = torch.autograd.grad(_pos_mlp_batched(self.rpe, Pm), [Pm[k] for k in _POS_KEYS], dpos, allow_unused=True)
        for k, gk in zip(_POS_KEYS, grads):
            for i, sa in enumerate(self.sas):
                p_ = dict(sa.pos.named_parameters())[k]
                put(p_, gk[i] if gk is not None else torch.zeros_like(p_))


def _pos_groups(m):
    groups: Dict[tuple, list] = {}
    for layer in m.layers:
        for blk in layer.blocks:
            if hasattr(blk.attn, "attns"):
                for sa in blk.attn.attns:
                    groups.setdefault((sa.num_heads, sa.H_sp * sa.W_sp, sa.rpe_biases.shape[0], sa.pos.pos_dim), []).append(sa)
    return [_PosBiasGroup(v) for v in groups.values()]


def _dense_bias(sa) -> torch.Tensor:
    """DynamicPosBias MLP on the offset table, gathered into [heads][N][N] (dat_arch.py:219-224)"""
    pos = sa.pos(sa.rpe_biases.float())
    N = sa.H_sp * sa.W_sp
    return pos[sa.relative_position_index.reshape(-1)].reshape(N, N, -1).permute(2, 0, 1).float().contiguous()


# ---- packed operands of the backward pass -----------------------------------------------------------------------------------------------
def pack_train(m, device) -> Dict[str, torch.Tensor]:
    ha = _ha()
    ver = sum(p._version for p in m.parameters())
    if getattr(m, "_packedT", None) is not None and m._packedT_version == ver and m._packedT_device == device:
        return m._packedT
    C_, CP = m.embed_dim, ha._rup(m.embed_dim, 64)
    hid = int(C_ * m.expansion_factor)
    half = hid // 2
    HPh = ha._rup(half, 64)
    P: Dict[str, torch.Tensor] = {}
    with torch.no_grad(), ha.batched_pack() as pk:
        rows = _da()._fc1_rows(hid, half, HPh, device)
        P["fc1_rows"] = rows
        for li, layer in enumerate(m.layers):
            nH = m.heads[li]
            dh, CA = C_ // nH, nH * 32
            hm = ha._head_map(nH, dh, device)
            qkv_rows = ha._qkv_rows(nH, dh, device)
            for bi, blk in enumerate(layer.blocks):
                pre = f"{li}.{bi}."
                at = blk.attn
                P[pre + "WqkvT"] = ha._pack_linear(at.qkv.weight.t(), CP, 3 * CA, col_map=qkv_rows)
                P[pre + "WprojT"] = ha._pack_linear(at.proj.weight.t(), CA, CP, row_map=hm)
                P[pre + "W1T"] = ha._pack_linear(blk.ffn.fc1.weight.t(), CP, 2 * HPh, col_map=rows)
                P[pre + "W2T"] = ha._pack_linear(blk.ffn.fc2.weight.t(), HPh, CP)
                w9 = torch.zeros(CA, 9, device=device)
                w9[hm] = at.dwconv[0].weight.float().reshape(C_, 9)
                P[pre + "dw_wf"] = w9.flip(1).contiguous()                         # the depth-wise dgrad: the same conv with flipped taps
                P[pre + "dw_b"] = ha._pack_vec(at.dwconv[0].bias, CA, row_map=hm)
                P[pre + "dw_gam"] = ha._pack_vec(at.dwconv[1].weight, CA, row_map=hm)          # BatchNorm affine in the head-padded layout
                P[pre + "dw_bet"] = ha._pack_vec(at.dwconv[1].bias, CA, row_map=hm)
                sg9 = torch.zeros(HPh, 9, device=device)
                sg9[:half] = blk.ffn.sg.conv.weight.float().reshape(half, 9)
                P[pre + "sg_wf"] = sg9.flip(1).contiguous()
                S2 = at.spatial_interaction[0].weight.shape[0]
                w0 = torch.zeros(S2, CA, device=device)
                w0[:, hm] = at.spatial_interaction[0].weight.float().reshape(S2, C_)
                P[pre + "si_w0raw"] = w0.contiguous()
            P[f"{li}.WconvT"] = _pack_conv_T(layer.conv.weight, CP, CP)
        P["WcabT"] = _pack_conv_T(m.conv_after_body.weight, CP, CP)
        if m.upsampler == 'pixelshuffle':
            P["WbeforeT"] = _pack_conv_T(m.conv_before_upsample[0].weight, CP, 64)
            k = 0
            for mod in m.upsample:
                if isinstance(mod, nn.Conv2d):
                    r = int(round(math.sqrt(mod.weight.shape[0] // 64)))
                    pm = ha._ps_map(mod.weight.shape[0], r, 64, device)
                    P[f"WupT{k}"] = _pack_conv_T(mod.weight, 64, mod.weight.shape[0], col_map=pm)
                    k += 1
        P["ones"] = torch.ones(max(2 * HPh, 256), device=device)
        P["zeros"] = torch.zeros(max(2 * HPh, 256), device=device)
        pk.resolve(P)
    m._packedT, m._packedT_version, m._packedT_device = P, ver, device
    return P


# ---- forward, keeping activations ---------------------------------------------------------------------------------------------------------
def dat_forward_train(m, x: torch.Tensor, P: Dict[str, torch.Tensor], PT: Dict[str, torch.Tensor], drop: Optional[torch.Tensor]):
    """drop: None or fp32 [n_blocks][2][B] DropPath factors (0 or 1 / keep_prob) of each block's attention and FFN branch."""
    ha = _ha()
    _gemm, _rup = ha._gemm, ha._rup
    dev = x.device
    st = torch.cuda.current_stream(dev).cuda_stream
    B, Cin, H, W = x.shape
    s0, s1 = m.split_size
    big = max(s0, s1)
    Hp, Wp = _rup(H, big), _rup(W, big)
    T, HW, s = B * H * W, H * W, m.upscale
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    half = int(C_ * m.expansion_factor) // 2
    HPh = _rup(half, 64)
    f32, b16 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
    L = lib()
    S: dict = dict(B=B, Cin=Cin, H=H, W=W, Hp=Hp, Wp=Wp, T=T, blocks=[], layers=[], drop=drop)
    ones, zeros = PT["ones"], PT["zeros"]

    mean3 = (C.c_float * 3)(*(m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]))
    img4 = torch.empty(T, 4, **f32)
    check(L.srk_img_prep(x.data_ptr(), img4.data_ptr(), B, Cin, H, W, H, W, float(m.img_range), C.byref(mean3), st))
    f0 = torch.empty(T, CP, **f32)
    check(L.srk_stem_conv(img4.data_ptr(), m.conv_first.weight.data_ptr(), m.conv_first.bias.data_ptr(), f0.data_ptr(), B, H, W, Cin, C_, CP, st))
    _, cur, mean_pe, rstd_pe = ops.layernorm_fwd(f0, m.before_RG[1].weight, m.before_RG[1].bias, C_, out_bf16=False, out_f32=True)
    S.update(img4=img4, f0=f0, mean_pe=mean_pe, rstd_pe=rstd_pe)

    # the dense position biases of every spatial block, from one batched evaluation of the position-bias MLPs
    pos_groups = _pos_groups(m)
    pos_of = {}
    for gq in pos_groups:
        dense = gq.forward()
        gq.d_bias = None
        for sa in gq.sas:
            pos_of[id(sa)] = (gq, gq.index[id(sa)])
    S["pos_groups"], S["pos_of"] = pos_groups, pos_of

    n_chunks = int(L.srk_chan_stats_chunks(HW))

    def token_sums(p, ldp, q, ldq, C8, per_sample=False):
        """-> (sum p, sum p q) over all tokens [8 C8] or per sample [B][8 C8]"""
        part = torch.empty(B, n_chunks, 2, C8 * 8, **f32)
        check(L.srk_chan_stats(p, ldp, q, ldq, part.data_ptr(), B, HW, C8, st))
        r = _sum_rows(part, B, n_chunks).view(B, 2, C8 * 8) if per_sample else _sum_rows(part, 1, B * n_chunks).view(2, C8 * 8)
        return r[..., 0, :], r[..., 1, :]

    bidx = 0
    for li, layer in enumerate(m.layers):
        nH = m.heads[li]
        dh, CA, hb = C_ // nH, nH * 32, nH // 2
        hm = ha._head_map(nH, dh, dev)
        scale = float(m.qk_scale or dh ** -0.5)
        layer_in = cur
        for bi, blk in enumerate(layer.blocks):
            pre = f"{li}.{bi}."
            at = blk.attn
            bk: dict = dict(li=li, bi=bi, pre=pre, blk=blk, nH=nH, CA=CA, scale=scale, bidx=bidx, x_in=cur, spatial=(bi % 2 == 0))
            xn1, _, bk["mean1"], bk["rstd1"] = ops.layernorm_fwd(cur, blk.norm1.weight, blk.norm1.bias, C_)
            qkv = torch.empty(T, 3 * CA, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, xn1, P[pre + "Wqkv"], T, 3 * CA, CP, lda=CP, bias=P[pre + "bqkv"], outb=qkv, ldo=3 * CA)
            v_ptr = qkv.data_ptr() + 2 * CA * 2
            # DW-conv branch on v (:418 / :508): conv (+ bias), BatchNorm with the batch's statistics, GELU
            c_pre, conv = torch.empty(T, CA, **b16), torch.empty(T, CA, **b16)
            check(L.srk_dwconv3x3(v_ptr, 3 * CA, P[pre + "dw_w"].data_ptr(), ones.data_ptr(), PT[pre + "dw_b"].data_ptr(), None, 0, c_pre.data_ptr(),
                                  CA, B, H, W, CA // 8, 0, st))
            part = torch.empty(B, n_chunks, 2, CA, **f32)
            check(L.srk_chan_stats(c_pre.data_ptr(), CA, c_pre.data_ptr(), CA, part.data_ptr(), B, HW, CA // 8, st))
            real_of = ha._cached_map(("real_of", nH, dh, str(dev)), lambda: torch.full((CA,), -1, dtype=torch.int32, device=dev).scatter_(
                0, hm, torch.arange(C_, dtype=torch.int32, device=dev)))
            dw_coef = _bn_forward(L, st, part, B * n_chunks, 2 * CA, CA, CA, T, PT[pre + "dw_gam"], PT[pre + "dw_bet"], at.dwconv[1], real_of)
            dw_s, dw_t = dw_coef[0], dw_coef[1]
            check(L.srk_affine_act_bf16(c_pre.data_ptr(), CA, dw_s.data_ptr(), dw_t.data_ptr(), conv.data_ptr(), CA, T, CA // 8, 0, 1, st))
            bk.update(dw_s=dw_s, dw_t=dw_t, dw_coef=dw_coef)
            att = torch.empty(T, CA, **b16)
            if bk["spatial"]:
                biases = []
                for br, (hs, wsz) in enumerate(((s0, s1), (s1, s0))):
                    sy, sx = (hs // 2, wsz // 2) if at.shifted else (0, 0)
                    off = br * hb * 32 * 2
                    gq, gi = pos_of[id(at.attns[br])]
                    bias = gq.dense[gi]
                    biases.append(bias)
                    check(L.srk_win_attention_fwd_padded(qkv.data_ptr() + off, 3 * CA, CA, bias.data_ptr(), 0, att.data_ptr() + off, CA, B, H, W, Hp,
                                                         Wp, hs, wsz, sy, sx, hb, scale, 0, st))
                bk["biases"] = biases
                gate_src, tok_src = conv, att            # channel map from the conv branch, spatial map from the attention
            else:
                part = torch.empty(int(L.srk_chan_gram_floats(B, HW, nH)), **f32)
                check(L.srk_chan_gram(qkv.data_ptr(), 3 * CA, qkv.data_ptr() + CA * 2, 3 * CA, part.data_ptr(), B, HW, nH, st))
                gram, A = torch.empty(B, nH, 1088, **f32), torch.empty(B, nH, 32, 32, **f32)      # G | sum q^2 | sum k^2 ; softmax(...)
                temp = at.temperature.float().contiguous()
                check(L.srk_chan_attn_matrix_fwd(part.data_ptr(), part.numel() // (B * nH * 1088), temp.data_ptr(), gram.data_ptr(), A.data_ptr(), B,
                                                 nH, dh, st))
                check(L.srk_chan_apply_mat(A.data_ptr(), v_ptr, 3 * CA, None, None, 0, att.data_ptr(), CA, B, HW, nH, 0, st))
                bk.update(gram=gram, A=A)
                gate_src, tok_src = att, conv            # channel map from the attention, spatial map from the conv branch
            # channel interaction on the pooled map (:315-321): [B][C] -> sigmoid gate
            pooled, _ = token_sums(gate_src.data_ptr(), CA, gate_src.data_ptr(), CA, CA // 8, per_sample=True)
            ci = at.channel_interaction
            S1 = ci[1].weight.shape[0]
            if _ci_fused_ok(B, C_, S1, ci):
                pm, cgate = torch.empty(B, C_, **f32), torch.empty(B, CA, **f32)
                bnc = ci[2]
                track = bnc.track_running_stats and bnc.running_mean is not None
                check(L.srk_channel_interaction_fwd(pooled.data_ptr(), pooled.stride(0), 1.0 / HW, _hm32(hm).data_ptr(), ci[1].weight.data_ptr(), ci[1].bias.data_ptr(),
                                                    bnc.weight.data_ptr(), bnc.bias.data_ptr(), float(bnc.eps), ci[4].weight.data_ptr(),
                                                    ci[4].bias.data_ptr(), bnc.running_mean.data_ptr() if track else None,
                                                    bnc.running_var.data_ptr() if track else None, float(bnc.momentum) if track else 0.0,
                                                    pm.data_ptr(), cgate.data_ptr(), B, C_, S1, CA, st))
                if track:
                    bnc.num_batches_tracked += 1
            else:
                pm = (pooled / HW)[:, hm].contiguous()
                cst: dict = {}
                cg = _channel_interaction(pm, ci, cst)
                _bn_update(ci[2], cst["mean"], cst["var"], cst["n"])
                cgate = torch.zeros(B, CA, **f32)
                cgate[:, hm] = cg
            # spatial interaction (:322-327): 1x1 conv -> BatchNorm (batch statistics) -> GELU -> 1x1 conv -> sigmoid gate per token
            si = at.spatial_interaction
            S2 = si[0].weight.shape[0]
            nblk = (T + 255) // 256
            part = torch.empty(nblk, 2, 16, **f32)
            w0raw, b0raw = PT[pre + "si_w0raw"], si[0].bias.float().contiguous()
            check(L.srk_spatial_gate_train(0, tok_src.data_ptr(), CA, w0raw.data_ptr(), b0raw.data_ptr(), None, None, None, None, None, None, None,
                                           None, 0, 0, part.data_ptr(), T, CA, S2, st))
            si_coef = _bn_forward(L, st, part, nblk, 32, 16, S2, T, si[1].weight.float(), si[1].bias.float(), si[1], None)
            si_s, si_t = si_coef[0, :S2], si_coef[1, :S2]
            tgate = torch.empty(T, **f32)
            w3 = si[3].weight.float().reshape(S2).contiguous()
            w0f, b0f = (w0raw * si_s[:, None]).contiguous(), (b0raw * si_s + si_t).contiguous()          # BatchNorm folded for the forward kernel
            b3 = si[3].bias.float().contiguous()
            check(L.srk_spatial_gate_dev(tok_src.data_ptr(), CA, w0f.data_ptr(), b0f.data_ptr(), w3.data_ptr(), b3.data_ptr(), S2, tgate.data_ptr(), T,
                                         CA, st))
            comb = torch.empty(T, CA, **b16)            # tok_src * cgate + gate_src * tgate  (:430-436 / :518-524)
            check(L.srk_dual_gate_combine(tok_src.data_ptr(), gate_src.data_ptr(), cgate.data_ptr(), tgate.data_ptr(), comb.data_ptr(), T, HW, CA, 0, st))
            x1 = torch.empty(T, CP, **f32)
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, comb, P[pre + "Wproj"], T, CP, CA, lda=CA, bias=P[pre + "bproj"], res=cur, outf=x1,
                  rowscale=None if drop is None else drop[bidx, 0], rows_per_sample=HW)
            xn2, _, bk["mean2"], bk["rstd2"] = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, C_)
            # SGFN (:74-90)
            u, hh = torch.empty(T, 2 * HPh, **b16), torch.empty(T, 2 * HPh, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_GELU, xn2, P[pre + "W1"], T, 2 * HPh, CP, lda=CP, bias=P[pre + "b1"], outb=u, outb2=hh)
            x2n, gated = torch.empty(T, HPh, **b16), torch.empty(T, HPh, **b16)
            sgm = blk.ffn.sg
            check(L.srk_rowln_bf16(hh.data_ptr() + HPh * 2, 2 * HPh, sgm.norm.weight.data_ptr(), sgm.norm.bias.data_ptr(), x2n.data_ptr(), HPh, T, half,
                                   HPh, st))
            check(L.srk_dwconv3x3(x2n.data_ptr(), HPh, P[pre + "sg_w"].data_ptr(), P[pre + "sg_s"].data_ptr(), P[pre + "sg_t"].data_ptr(), hh.data_ptr(),
                                  2 * HPh, gated.data_ptr(), HPh, B, H, W, HPh // 8, 0, st))
            nxt = torch.empty(T, CP, **f32)
            last = bi == len(layer.blocks) - 1
            xb = torch.empty(T, CP, **b16) if last else None
            _gemm(st, _lib.LD_ROWS, _lib.EP_RES, gated, P[pre + "W2"], T, CP, HPh, lda=HPh, bias=P[pre + "b2"], res=x1, outf=nxt, outb=xb,
                  rowscale=None if drop is None else drop[bidx, 1], rows_per_sample=HW)
            bk.update(xn1=xn1, qkv=qkv, c_pre=c_pre, conv=conv, att=att, cgate=cgate, tgate=tgate, pm=pm, comb=comb, x1=x1, xn2=xn2, u=u, hh=hh,
                      x2n=x2n, gated=gated, si_coef=si_coef)
            S["blocks"].append(bk)
            cur = nxt
            bidx += 1
        nxt = torch.empty(T, CP, **f32)
        _gemm(st, _lib.LD_CONV3, _lib.EP_RES, xb, P[f"{li}.Wconv"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P[f"{li}.bconv"], res=layer_in, outf=nxt)
        S["layers"].append(dict(li=li, xb=xb, n_blocks=len(layer.blocks)))
        cur = nxt

    xnf, _, meanf, rstdf = ops.layernorm_fwd(cur, m.norm.weight, m.norm.bias, C_)
    fb = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_RES_BF16, xnf, P["Wcab"], T, CP, 9 * CP, conv=(B, H, W, CP), bias=P["bcab"], res=f0, outb=fb)
    S.update(x_last=cur, xnf=xnf, meanf=meanf, rstdf=rstdf, fb=fb, ups=[])
    y = torch.empty(B, Cin, H * s, W * s, **f32)
    mean4 = (m.mean.flatten().tolist() if m.in_chans == 3 else [0.0, 0.0, 0.0]) + [0.0]
    img = dict(inv_range=1.0 / float(m.img_range), Cimg=Cin, Hc=H * s, Wc=W * s, mean=mean4)
    if m.upsampler == 'pixelshuffle':
        t1 = torch.empty(T, 64, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_LRELU, fb, P["Wbefore"], T, 64, 9 * CP, conv=(B, H, W, CP), bias=P["bbefore"], outb=t1, scale=0.01)
        S["t1"] = t1
        src, h_, w_, k = t1, H, W, 0
        while f"Wup{k}" in P:
            r = int(P[f"rup{k}"])
            N = P[f"Wup{k}"].shape[0]
            up = torch.empty(B * h_ * r * w_ * r, 64, **b16)
            _gemm(st, _lib.LD_CONV3, _lib.EP_PS, src, P[f"Wup{k}"], B * h_ * w_, N, 9 * 64, conv=(B, h_, w_, 64), bias=P[f"bup{k}"], outb=up, r=r, Cs=64,
                  ldo=N)
            S["ups"].append(dict(src=src, out=up, h=h_, w=w_, r=r, N=N))
            src, h_, w_, k = up, h_ * r, w_ * r, k + 1
        _gemm(st, _lib.LD_CONV3, _lib.EP_IMG, src, P["Wlast"], B * h_ * w_, 16, 9 * 64, conv=(B, h_, w_, 64), bias=P["blast"], outf=y, img=img)
        S.update(hr_h=h_, hr_w=w_)
    else:
        _gemm(st, _lib.LD_CONV3, _lib.EP_PS_IMG, fb, P["Wdirect"], T, 16, 9 * CP, conv=(B, H, W, CP), bias=P["bdirect"], outf=y, img=dict(img), r=s)
    return y, S


# ---- backward ---------------------------------------------------------------------------------------------------------------------------------
def dat_backward(m, S: dict, dy: torch.Tensor, hook=None) -> Dict[str, torch.Tensor]:
    """-> {parameter name: gradient}.  hook (distributed.ListGradSynchronizer or None) gets each finished segment's gradients
    (tail, every ResidualGroup, head) so that their all-reduce overlaps the next segment."""
    ha = _ha()
    _gemm, _rup, _ptr = ha._gemm, ha._rup, ha._ptr
    dev = dy.device
    P = m._pack(dev, True)
    PT = pack_train(m, dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    B, Cin, H, W, Hp, Wp, T = S["B"], S["Cin"], S["H"], S["W"], S["Hp"], S["Wp"], S["T"]
    HW, s = H * W, m.upscale
    s0, s1 = m.split_size
    C_, CP = m.embed_dim, _rup(m.embed_dim, 64)
    hid = int(C_ * m.expansion_factor)
    half = hid // 2
    HPh = _rup(half, 64)
    f32, b16 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
    L = lib()
    drop = S["drop"]
    ones, zeros = PT["ones"], PT["zeros"]
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

    def put(p, g):
        G[pname(p)] = g.reshape(p.shape).contiguous()

    pending = []          # the block's linear weight gradients: queued, then ONE launch for all four (flush_wgrads)

    def lin_wgrad(y, x, lin, row_map=None, col_map=None):
        pending.append((y, x, lin, row_map, col_map))        # y and x must stay untouched until flush_wgrads

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
        dw, db = ops.zeros_f32((NP, 9 * CinP), dev), ops.zeros_f32((NP,), dev)
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
        check(L.srk_layernorm_bwd(dyb.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), norm.weight.data_ptr(), gx.data_ptr(), _ptr(gxb),
                                  dg.data_ptr(), dbt.data_ptr(), T, C_, CP, 1 if accumulate else 0, st))
        G[pname(norm.weight)], G[pname(norm.bias)] = dg, dbt

    def scaled(gb, bidx, which):
        if drop is None:
            return gb
        out = torch.empty_like(gb)
        check(L.srk_rowscale_bf16(gb.data_ptr(), out.data_ptr(), drop[bidx, which].data_ptr(), T, HW, CP, st))
        return out

    n_chunks = int(L.srk_chan_stats_chunks(HW))

    def token_sums(p, ldp, q, ldq, C8):
        part = torch.empty(B, n_chunks, 2, C8 * 8, **f32)
        check(L.srk_chan_stats(p, ldp, q, ldq, part.data_ptr(), B, HW, C8, st))
        r = _sum_rows(part, 1, B * n_chunks).view(2, C8 * 8)
        return r[0], r[1]

    def dwconv_grads(dyt, lddy, xptr, ldx, CPc, conv, real_rows):
        """depth-wise 3x3 weight / bias gradient from d y [T][lddy] and the conv's input -> parameter-shaped tensors"""
        nyb = int(L.srk_dwconv3x3_wgrad_chunks(H))
        part = torch.empty(B, nyb, 10, CPc, **f32)
        check(L.srk_dwconv3x3_wgrad(dyt, lddy, xptr, ldx, part.data_ptr(), B, H, W, CPc // 8, st))
        g = _sum_rows(part, 1, B * nyb).view(10, CPc)
        put(conv.weight, g[:9, real_rows].t())
        put(conv.bias, g[9, real_rows])

    # ---------------- reconstruction tail ----------------
    if m.upsampler == 'pixelshuffle':
        hs_, ws_ = S["hr_h"], S["hr_w"]
        gyimg = torch.empty(B * hs_ * ws_, 4, **f32)
        check(L.srk_img_grad_prep(dy.data_ptr(), gyimg.data_ptr(), B, Cin, H * s, W * s, hs_, ws_, 1, 4, 1.0 / float(m.img_range), st))
        last_in = S["ups"][-1]["out"] if S["ups"] else S["t1"]
        dwl, dbl = torch.zeros_like(m.conv_last.weight, dtype=torch.float32), torch.zeros_like(m.conv_last.bias, dtype=torch.float32)
        check(L.srk_smallconv_wgrad(last_in.data_ptr(), gyimg.data_ptr(), dwl.data_ptr(), dbl.data_ptr(), B, hs_, ws_, 64, 64, Cin, 4, st))
        G[pname(m.conv_last.weight)], G[pname(m.conv_last.bias)] = dwl, dbl
        gcur = torch.empty(B * hs_ * ws_, 64, **b16)
        check(L.srk_smallconv_dgrad(gyimg.data_ptr(), m.conv_last.weight.data_ptr(), gcur.data_ptr(), B, hs_, ws_, 64, 64, Cin, 4, st))
        up_convs = [mod for mod in m.upsample if isinstance(mod, nn.Conv2d)]
        for k in range(len(S["ups"]) - 1, -1, -1):
            u = S["ups"][k]
            r, N, h_, w_ = u["r"], u["N"], u["h"], u["w"]
            conv_wgrad(gcur, u["src"], up_convs[k], B, h_, w_, 64, N, r=r, row_map=ha._ps_map(N, r, 64, dev))
            gprev = torch.empty(B * h_ * w_, 64, **b16)
            if k == 0:
                _gemm(st, _lib.LD_CONV3_PS, _lib.EP_DLRELU, gcur, PT[f"WupT{k}"], B * h_ * w_, 64, 9 * N, conv=(B, h_, w_, N), r=r, Cs=64, outb=gprev,
                      aux=S["t1"], scale=0.01, ldo=64)
            else:
                _gemm(st, _lib.LD_CONV3_PS, _lib.EP_BF16, gcur, PT[f"WupT{k}"], B * h_ * w_, 64, 9 * N, conv=(B, h_, w_, N), r=r, Cs=64, outb=gprev, ldo=64)
            gcur = gprev
        conv_wgrad(gcur, S["fb"], m.conv_before_upsample[0], B, H, W, CP, 64)
        gfb = torch.empty(T, CP, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_BF16, gcur, PT["WbeforeT"], T, CP, 9 * 64, conv=(B, H, W, 64), outb=gfb)
    else:
        Co = s * s * Cin
        gyimg = torch.empty(T, 16, **f32)
        check(L.srk_img_grad_prep(dy.data_ptr(), gyimg.data_ptr(), B, Cin, H * s, W * s, H, W, s, 16, 1.0 / float(m.img_range), st))
        cv = m.upsample[0]
        dwl, dbl = torch.zeros_like(cv.weight, dtype=torch.float32), torch.zeros_like(cv.bias, dtype=torch.float32)
        check(L.srk_smallconv_wgrad(S["fb"].data_ptr(), gyimg.data_ptr(), dwl.data_ptr(), dbl.data_ptr(), B, H, W, C_, CP, Co, 16, st))
        G[pname(cv.weight)], G[pname(cv.bias)] = dwl, dbl
        gfb = torch.empty(T, CP, **b16)
        check(L.srk_smallconv_dgrad(gyimg.data_ptr(), cv.weight.data_ptr(), gfb.data_ptr(), B, H, W, C_, CP, Co, 16, st))
    conv_wgrad(gfb, S["xnf"], m.conv_after_body, B, H, W, CP, CP)
    dxn = torch.empty(T, CP, **b16)
    _gemm(st, _lib.LD_CONV3, _lib.EP_BF16, gfb, PT["WcabT"], T, CP, 9 * CP, conv=(B, H, W, CP), outb=dxn)
    gx, gxb = torch.empty(T, CP, **f32), torch.empty(T, CP, **b16)      # gradient of the current layer's output (later: input)
    ln_bwd(dxn, S["x_last"], S["meanf"], S["rstdf"], m.norm, gx, gxb, accumulate=False)
    segment_done()

    # ---------------- layers, last to first ----------------
    blocks = S["blocks"]
    pos = len(blocks)
    attn_scratch = None
    for lay in reversed(S["layers"]):
        li = lay["li"]
        layer = m.layers[li]
        conv_wgrad(gxb, lay["xb"], layer.conv, B, H, W, CP, CP)
        gx2, gxb2 = torch.empty(T, CP, **f32), torch.empty(T, CP, **b16)
        _gemm(st, _lib.LD_CONV3, _lib.EP_F32_BF16, gxb, PT[f"{li}.WconvT"], T, CP, 9 * CP, conv=(B, H, W, CP), outf=gx2, outb=gxb2)
        for _ in range(lay["n_blocks"]):
            pos -= 1
            bk = blocks[pos]
            pre, blk, nH, CA, bidx = bk["pre"], bk["blk"], bk["nH"], bk["CA"], bk["bidx"]
            at = blk.attn
            dh, hb = C_ // nH, nH // 2
            hm = ha._head_map(nH, dh, dev)
            qkv_rows = ha._qkv_rows(nH, dh, dev)
            qkv = bk["qkv"]
            v_ptr = qkv.data_ptr() + 2 * CA * 2
            # ---- SGFN: x2 = x1 + f * fc2(x1h * dwconv(LN(x2h))) with (x1h | x2h) = gelu(fc1(norm2 x1)) ----
            g_mlp = scaled(gxb2, bidx, 1)
            lin_wgrad(g_mlp, bk["gated"], blk.ffn.fc2)
            dgated = torch.empty(T, HPh, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, g_mlp, PT[pre + "W2T"], T, HPh, CP, lda=CP, outb=dgated)
            hh, sgm = bk["hh"], blk.ffn.sg
            cx = torch.empty(T, HPh, **b16)                       # dwconv(LN(x2h)) + bias, recomputed
            check(L.srk_dwconv3x3(bk["x2n"].data_ptr(), HPh, P[pre + "sg_w"].data_ptr(), P[pre + "sg_s"].data_ptr(), P[pre + "sg_t"].data_ptr(), None, 0,
                                  cx.data_ptr(), HPh, B, H, W, HPh // 8, 0, st))
            dhh, dcx = torch.empty(T, 2 * HPh, **b16), torch.empty(T, HPh, **b16)
            check(L.srk_mul_bwd_bf16(dgated.data_ptr(), HPh, hh.data_ptr(), 2 * HPh, cx.data_ptr(), HPh, dhh.data_ptr(), 2 * HPh, dcx.data_ptr(), HPh, T,
                                     HPh // 8, st))
            dwconv_grads(dcx.data_ptr(), HPh, bk["x2n"].data_ptr(), HPh, HPh, sgm.conv, _arange(half, dev))
            dx2n = cx                                               # (re-used buffer)
            check(L.srk_dwconv3x3(dcx.data_ptr(), HPh, PT[pre + "sg_wf"].data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, 0, dx2n.data_ptr(), HPh, B,
                                  H, W, HPh // 8, 0, st))
            nb = int(L.srk_rowln_bwd_blocks(T))
            part = torch.empty(nb, 2, half, **f32)
            check(L.srk_rowln_bwd_bf16(dx2n.data_ptr(), HPh, hh.data_ptr() + HPh * 2, 2 * HPh, sgm.norm.weight.data_ptr(), dhh.data_ptr() + HPh * 2,
                                       2 * HPh, part.data_ptr(), T, half, HPh, st))
            ps = _sum_rows(part, 1, nb).view(2, half)
            put(sgm.norm.weight, ps[0])
            put(sgm.norm.bias, ps[1])
            du = torch.empty(T, 2 * HPh, **b16)
            check(L.srk_dgelu_affine_bf16(dhh.data_ptr(), 2 * HPh, bk["u"].data_ptr(), 2 * HPh, ones.data_ptr(), zeros.data_ptr(), du.data_ptr(), 2 * HPh,
                                          T, 2 * HPh // 8, st))
            lin_wgrad(du, bk["xn2"], blk.ffn.fc1, row_map=PT["fc1_rows"])
            dxn2 = torch.empty(T, CP, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, du, PT[pre + "W1T"], T, CP, 2 * HPh, lda=2 * HPh, outb=dxn2)
            g1b = torch.empty(T, CP, **b16)
            ln_bwd(dxn2, bk["x1"], bk["mean2"], bk["rstd2"], blk.norm2, gx2, g1b, accumulate=True)        # gx2 = d x1
            # ---- attention half: x1 = x + f * proj(tok_src * cgate + gate_src * tgate) ----
            g_att = scaled(g1b, bidx, 0)
            lin_wgrad(g_att, bk["comb"], at.proj, col_map=hm)
            dcomb = torch.empty(T, CA, **b16)
            _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, g_att, PT[pre + "WprojT"], T, CA, CP, lda=CP, outb=dcomb, ldo=CA)
            att, conv = bk["att"], bk["conv"]
            gate_src, tok_src = (conv, att) if bk["spatial"] else (att, conv)
            d_tok_src, d_gate_src = torch.empty(T, CA, **b16), torch.empty(T, CA, **b16)
            nck = (HW + 63) // 64
            dcg_part, dsmap = torch.empty(B, nck, CA, **f32), torch.empty(T, **f32)
            check(L.srk_dual_gate_bwd(dcomb.data_ptr(), tok_src.data_ptr(), gate_src.data_ptr(), bk["cgate"].data_ptr(), bk["tgate"].data_ptr(),
                                      d_tok_src.data_ptr(), d_gate_src.data_ptr(), dcg_part.data_ptr(), dsmap.data_ptr(), B, HW, CA, st))
            # channel interaction: autograd on the [B][C] function, then the pooled gradient broadcast back over the tokens
            ci = at.channel_interaction
            ci_params = [ci[1].weight, ci[1].bias, ci[2].weight, ci[2].bias, ci[4].weight, ci[4].bias]
            S1 = ci[1].weight.shape[0]
            if _ci_fused_ok(B, C_, S1, ci):
                dcg = _sum_rows(dcg_part, B, nck)                                       # [B][CA], head-padded
                gci = [torch.empty(p_.shape, **f32) for p_ in ci_params]
                dpool = torch.empty(B, CA, **f32)
                check(L.srk_channel_interaction_bwd(bk["pm"].data_ptr(), dcg.data_ptr(), CA, 1.0 / HW, _hm32(hm).data_ptr(), ci[1].weight.data_ptr(),
                                                    ci[1].bias.data_ptr(), ci[2].weight.data_ptr(), ci[2].bias.data_ptr(), float(ci[2].eps),
                                                    ci[4].weight.data_ptr(), ci[4].bias.data_ptr(), gci[0].data_ptr(), gci[1].data_ptr(),
                                                    gci[2].data_ptr(), gci[3].data_ptr(), gci[4].data_ptr(), gci[5].data_ptr(), dpool.data_ptr(), B, C_,
                                                    S1, CA, st))
                for p_, g_ in zip(ci_params, gci):
                    G[pname(p_)] = g_
            else:
                with torch.enable_grad():
                    pm = bk["pm"].detach().requires_grad_(True)
                    cg = _channel_interaction(pm, ci)
                    grads = torch.autograd.grad(cg, [pm] + ci_params, dcg_part.sum(1)[:, hm])
                for p_, g_ in zip(ci_params, grads[1:]):
                    put(p_, g_)
                dpool = torch.zeros(B, CA, **f32)
                dpool[:, hm] = grads[0] / HW
            check(L.srk_lincomb2_bf16(None, 0, None, 0, None, None, dpool.data_ptr(), d_gate_src.data_ptr(), CA, T, CA // 8, HW, 1, st))
            # spatial interaction backward (closed-form BatchNorm backward between two token passes)
            si = at.spatial_interaction
            S2 = si[0].weight.shape[0]
            nblk = (T + 255) // 256
            w0raw, b0raw = PT[pre + "si_w0raw"], si[0].bias.float().contiguous()
            w3 = si[3].weight.float().reshape(S2).contiguous()
            si_coef = bk["si_coef"]
            si_s, si_t = si_coef[0], si_coef[1]              # 16-entry rows, the first S2 valid
            part = torch.empty(nblk, 4, 16, **f32)
            check(L.srk_spatial_gate_train(1, tok_src.data_ptr(), CA, w0raw.data_ptr(), b0raw.data_ptr(), si_s.data_ptr(), si_t.data_ptr(), w3.data_ptr(),
                                           dsmap.data_ptr(), None, None, None, None, 0, 0, part.data_ptr(), T, CA, S2, st))
            ps = _sum_rows(part, 1, nblk).view(4, 16)
            bc = torch.empty(5, 16, **f32)
            check(L.srk_bn_train_bwd_coeffs(part.data_ptr(), nblk, 64, 16, S2, float(T), si_coef.data_ptr(), bc.data_ptr(), st))
            cA, cB, cC = bc[0], bc[1], bc[2]
            put(si[1].weight, bc[3, :S2])
            put(si[1].bias, bc[4, :S2])
            put(si[3].weight, ps[2, :S2])
            put(si[3].bias, ps[3, :1])
            part = torch.empty(nblk, 16, CA + 1, **f32)
            check(L.srk_spatial_gate_train(2, tok_src.data_ptr(), CA, w0raw.data_ptr(), b0raw.data_ptr(), si_s.data_ptr(), si_t.data_ptr(), w3.data_ptr(),
                                           dsmap.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr(), d_tok_src.data_ptr(), CA, 1, part.data_ptr(), T,
                                           CA, S2, st))
            ps = _sum_rows(part, 1, nblk).view(-1)
            put(si[0].weight, ps[:16 * CA].view(16, CA)[:S2][:, hm])
            put(si[0].bias, ps[16 * CA:16 * CA + S2])
            d_att, d_conv = (d_tok_src, d_gate_src) if bk["spatial"] else (d_gate_src, d_tok_src)
            # ---- DW-conv branch: conv = gelu(BatchNorm(dwconv(v) + b)) ----
            c_pre = bk["c_pre"]
            dz = torch.empty(T, CA, **b16)
            check(L.srk_dgelu_affine_bf16(d_conv.data_ptr(), CA, c_pre.data_ptr(), CA, bk["dw_s"].data_ptr(), bk["dw_t"].data_ptr(), dz.data_ptr(), CA, T,
                                          CA // 8, st))
            part = torch.empty(B, n_chunks, 2, CA, **f32)
            check(L.srk_chan_stats(dz.data_ptr(), CA, c_pre.data_ptr(), CA, part.data_ptr(), B, HW, CA // 8, st))
            bc = torch.empty(5, CA, **f32)
            check(L.srk_bn_train_bwd_coeffs(part.data_ptr(), B * n_chunks, 2 * CA, CA, CA, float(T), bk["dw_coef"].data_ptr(), bc.data_ptr(), st))
            cA, cB, cC = bc[0], bc[1], bc[2]
            put(at.dwconv[1].weight, bc[3][hm])
            put(at.dwconv[1].bias, bc[4][hm])
            dcpre = d_conv                                          # (re-used buffer)
            check(L.srk_lincomb2_bf16(dz.data_ptr(), CA, c_pre.data_ptr(), CA, cA.data_ptr(), cB.data_ptr(), cC.data_ptr(), dcpre.data_ptr(), CA, T,
                                      CA // 8, 0, 0, st))
            dwconv_grads(dcpre.data_ptr(), CA, v_ptr, 3 * CA, CA, at.dwconv[0], hm)
            dv_conv = dz                                            # (re-used buffer)
            check(L.srk_dwconv3x3(dcpre.data_ptr(), CA, PT[pre + "dw_wf"].data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, 0, dv_conv.data_ptr(), CA,
                                  B, H, W, CA // 8, 0, st))
            # ---- attention core ----
            dqkv = torch.empty(T, 3 * CA, **b16) if not _POISON else torch.full((T, 3 * CA), float("nan"), **b16)      # every element is written by the attention backward
            if bk["spatial"]:
                for br, (hs, wsz) in enumerate(((s0, s1), (s1, s0))):
                    sa = at.attns[br]
                    sy, sx = (hs // 2, wsz // 2) if at.shifted else (0, 0)
                    off = br * hb * 32 * 2
                    N = hs * wsz
                    gq, gi = S["pos_of"][id(sa)]
                    if gq.d_bias is None:
                        gq.d_bias = torch.zeros_like(gq.dense)
                    dbias = gq.d_bias[gi]                                 # accumulated in place; the MLPs' backward runs once, at the end
                    need = int(L.srk_win_attention_bwd_padded_scratch(B, Hp, Wp, hs, wsz, hb))
                    if attn_scratch is None or attn_scratch.numel() < need:
                        attn_scratch = torch.empty(need, dtype=torch.uint8, device=dev)
                    check(L.srk_win_attention_bwd_padded(qkv.data_ptr() + off, 3 * CA, CA, bk["biases"][br].data_ptr(), d_att.data_ptr() + off, CA,
                                                         dqkv.data_ptr() + off, dbias.data_ptr(), attn_scratch.data_ptr(), B, H, W, Hp, Wp, hs, wsz, sy,
                                                         sx, hb, bk["scale"], st))
            else:
                part = torch.empty(int(L.srk_chan_gram_floats(B, HW, nH)), **f32)
                check(L.srk_chan_gram(d_att.data_ptr(), CA, v_ptr, 3 * CA, part.data_ptr(), B, HW, nH, st))
                At = bk["A"].transpose(-1, -2).contiguous()
                check(L.srk_chan_apply_mat(At.data_ptr(), d_att.data_ptr(), CA, None, None, 0, dqkv.data_ptr() + 2 * CA * 2, 3 * CA, B, HW, nH, 0, st))
                dG, dGt = torch.empty(B, nH, 32, 32, **f32), torch.empty(B, nH, 32, 32, **f32)
                dsq2, dsk2, dtemp = torch.empty(B, nH, 32, **f32), torch.empty(B, nH, 32, **f32), torch.empty(B, nH, **f32)
                temp = at.temperature.float().contiguous()
                check(L.srk_chan_attn_matrix_bwd(part.data_ptr(), part.numel() // (B * nH * 1088), bk["gram"].data_ptr(), bk["A"].data_ptr(),
                                                 temp.data_ptr(), dG.data_ptr(), dGt.data_ptr(), dsq2.data_ptr(), dsk2.data_ptr(), dtemp.data_ptr(), B, nH,
                                                 dh, st))
                put(at.temperature, _sum_rows(dtemp, 1, B).view(-1))
                q_ptr, k_ptr = qkv.data_ptr(), qkv.data_ptr() + CA * 2
                check(L.srk_chan_apply_mat(dG.data_ptr(), k_ptr, 3 * CA, dsq2.data_ptr(), q_ptr, 3 * CA, dqkv.data_ptr(), 3 * CA, B, HW, nH, 0, st))
                check(L.srk_chan_apply_mat(dGt.data_ptr(), q_ptr, 3 * CA, dsk2.data_ptr(), k_ptr, 3 * CA, dqkv.data_ptr() + CA * 2, 3 * CA, B, HW, nH, 0, st))
            check(L.srk_lincomb2_bf16(dv_conv.data_ptr(), CA, None, 0, None, None, None, dqkv.data_ptr() + 2 * CA * 2, 3 * CA, T, CA // 8, 0, 1, st))
            lin_wgrad(dqkv, bk["xn1"], at.qkv, row_map=qkv_rows)
            flush_wgrads()            # before the kernel below overwrites gxb2 (the fc2 gradient's operand when no DropPath copy was made)
            if CP in (64, 128, 192):      # qkv dgrad with the norm1 backward in its epilogue (gx2 += d x, gxb2 = its bf16 copy)
                dg, dbt = ops.zeros_f32((C_,), dev), ops.zeros_f32((C_,), dev)
                _gemm(st, _lib.LD_ROWS, _lib.EP_LNBWD, dqkv, PT[pre + "WqkvT"], T, CP, 3 * CA, lda=3 * CA, outf=gx2, outb=gxb2, ldo=CP,
                      ln=dict(x=bk["x_in"], mean=bk["mean1"], rstd=bk["rstd1"], gamma=blk.norm1.weight, dgamma=dg, dbeta=dbt, C=C_))
                G[pname(blk.norm1.weight)], G[pname(blk.norm1.bias)] = dg, dbt
            else:
                dxn1 = torch.empty(T, CP, **b16)
                _gemm(st, _lib.LD_ROWS, _lib.EP_BF16, dqkv, PT[pre + "WqkvT"], T, CP, 3 * CA, lda=3 * CA, outb=dxn1)
                ln_bwd(dxn1, bk["x_in"], bk["mean1"], bk["rstd1"], blk.norm1, gx2, gxb2, accumulate=True)
        check(L.srk_add_f32_bf16(gx.data_ptr(), gx2.data_ptr(), gxb.data_ptr(), T * CP, st))      # d(layer input) = d(body input) + d(layer output)
        segment_done()

    for gq in S["pos_groups"]:            # every position-bias MLP's parameter gradients from one batched backward
        if gq.d_bias is not None:
            gq.backward(put)
    # ---------------- head: before_RG's LayerNorm, long skip, conv_first ----------------
    gf = torch.empty(T, CP, **f32)
    ln_bwd(gxb, S["f0"], S["mean_pe"], S["rstd_pe"], m.before_RG[1], gf, None, accumulate=False)
    check(L.srk_add_bf16_into_f32(gf.data_ptr(), gfb.data_ptr(), T * CP, st))
    dwf, dbf = torch.zeros_like(m.conv_first.weight, dtype=torch.float32), torch.zeros_like(m.conv_first.bias, dtype=torch.float32)
    check(L.srk_stem_wgrad(S["img4"].data_ptr(), gf.data_ptr(), dwf.data_ptr(), dbf.data_ptr(), B, H, W, Cin, C_, CP, st))
    G[pname(m.conv_first.weight)], G[pname(m.conv_first.bias)] = dwf, dbf
    segment_done()
    if hook is not None:
        hook.finish()
    return G


class DATFunction(torch.autograd.Function):
    """One autograd node for the whole model (as HATFunction): forward keeps the activations and updates the BatchNorm running
    statistics, backward returns every parameter's gradient.  The input image gets no gradient."""

    @staticmethod
    def forward(ctx, model, x, drop, *params):
        with torch.cuda.device(x.device), torch.no_grad():
            y, saved = dat_forward_train(model, x.contiguous().float(), model._pack(x.device, True), pack_train(model, x.device), drop)
        ctx.model, ctx.saved = model, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        model = ctx.model
        arena = model.__dict__.setdefault("_zero_arena", ops.ZeroArena())      # the pass's zeroed accumulators: one buffer, one fill
        with torch.cuda.device(dy.device), ops.arena_scope(arena, dy.device):
            G = dat_backward(model, ctx.saved, dy.contiguous().float(), hook=getattr(model, "grad_sync", None))
        ctx.saved = None
        grads = []
        for n, p in model.named_parameters():
            g = G.get(n)
            grads.append(None if g is None else g.reshape(p.shape).to(p.dtype))
        return (None, None, None, *grads)
