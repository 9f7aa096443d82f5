"""The re-projecting attention backward (csrc/attn_bwd_fused.hip, through srk_window_attention_bwd_fused) against

  (1) plain fp32 math of network_swinir.py:121-143 backwards (autograd on a restatement that rounds q, k, v and d attn_out to
      bf16 where the kernel does), and
  (2) the composition it replaces: q/k/v and d attn_out materialised in memory + srk_window_attention_bwd (csrc/attn.hip),
      which the whole-model gradient goldens pin.
"""
import pytest
import torch

from oracle import swinir_oracle as O
from test_gpu_kernels import attention_ref, bf, close_bf16, dev, pad_qkv

pytestmark = pytest.mark.gpu

C, CP, NH, D = 180, 192, 6, 30


@pytest.fixture(scope="module")
def ops():
    from tpu_superresolution_amd import ops as _ops
    return _ops


def packed_operands(B_, seed):
    g = torch.Generator().manual_seed(seed)
    xn = torch.zeros(B_ * 64, CP)
    xn[:, :C] = torch.randn(B_ * 64, C, generator=g)
    dx1 = torch.zeros(B_ * 64, CP)
    dx1[:, :C] = torch.randn(B_ * 64, C, generator=g) * 0.5
    wqkv = torch.randn(3, NH, D, C, generator=g) * 0.08          # [which][head][d][c]
    bqkv = torch.randn(3, NH, D, generator=g) * 0.2
    wproj = torch.randn(C, NH, D, generator=g) * 0.08             # proj.weight[c_out][head, d]
    table = torch.randn(225, NH, generator=g) * 0.5
    # packed layouts of the kernels: head_dim 30 -> 32, C 180 -> 192, zero padding
    wq_p = torch.zeros(3, NH, 32, CP)
    wq_p[:, :, :D, :C] = wqkv
    bq_p = torch.zeros(3, NH, 32)
    bq_p[:, :, :D] = bqkv
    wpt_p = torch.zeros(NH, 32, CP)                               # [attention channel (head, d)][output channel]
    wpt_p[:, :D, :C] = wproj.permute(1, 2, 0)
    return (bf(xn), bf(dx1), bf(wq_p.reshape(3 * NH * 32, CP)), bq_p.reshape(-1).contiguous(), bf(wpt_p.reshape(NH * 32, CP)), table)


def materialise(xn, dx1, wq, bq, wpt, scale):
    """q (scaled), k, v and d attn_out as the kernels materialise them: fp32 accumulate, then bf16."""
    B_ = xn.shape[0] // 64
    qkv = (xn.float() @ wq.float().t() + bq).view(B_, 64, 3, NH, 32).permute(2, 0, 3, 1, 4)    # [3, B_, nH, 64, 32]
    q, k, v = bf(qkv[0] * scale).float(), bf(qkv[1]).float(), bf(qkv[2]).float()
    dao = bf(dx1.float() @ wpt.float().t())                                                      # [B_*64, nH*32]
    return q, k, v, dao


@pytest.mark.parametrize("B,H,W,shift", [(4, 64, 64, 0), (4, 64, 64, 4), (5, 64, 64, 4), (16, 32, 40, 4)])
def test_fused_attention_backward_vs_fp32_math_and_vs_the_separate_kernels(ops, B, H, W, shift):
    B_ = B * (H // 8) * (W // 8)
    xn, dx1, wq, bq, wpt, table = packed_operands(B_, 11 + shift)
    scale = D ** -0.5
    biasd = ops.rel_pos_bias_expand(dev(table))
    dqkv, dtab = ops.window_attention_bwd_fused(dev(xn), dev(wq), dev(bq), scale, dev(dx1), dev(wpt), biasd, H, W, shift)
    got = dqkv.float().cpu().view(B_, 64, 3, NH, 32).permute(2, 0, 3, 1, 4)                     # [3, B_, nH, 64, 32]

    q, k, v, dao = materialise(xn, dx1, wq, bq, wpt, scale)
    # (2) the composition it replaces
    dqkv2, dtab2 = ops.window_attention_bwd(dev(bf(torch.stack([q, k, v]))), biasd, dev(dao), scale, H, W, shift)
    ref2 = dqkv2.float().cpu().view(B_, 64, 3, NH, 32).permute(2, 0, 3, 1, 4)
    for i in range(3):
        close_bf16(got[i], ref2[i], 1e-2 * float(ref2[i].abs().max()), rtol=2e-2)
    assert (dtab.cpu() - dtab2.cpu()).abs().max() < 5e-3 * max(1.0, float(dtab2.abs().max()))

    # (1) fp32 math on the same bf16-rounded q / k / v / d attn_out
    qr = (q[..., :D] / scale).requires_grad_(True)
    kr, vr, tr = k[..., :D].clone().requires_grad_(True), v[..., :D].clone().requires_grad_(True), table.clone().requires_grad_(True)
    o, _ = attention_ref(qr * scale, kr, vr, tr, H, W, shift)
    o.backward(dao.float().view(B_, 64, NH, 32).permute(0, 2, 1, 3)[..., :D])
    for i, ref in enumerate((qr.grad, kr.grad, vr.grad)):
        close_bf16(got[i][..., :D], ref, 2e-2 * float(ref.abs().max()), rtol=2e-2)
        assert (got[i][..., D:] == 0).all()
    assert (dtab.cpu() - tr.grad).abs().max() < 2e-2 * max(1.0, float(tr.grad.abs().max()))


def test_fused_attention_backward_refuses_other_widths(ops):
    from tpu_superresolution_amd._lib import SrkUnsupported
    xn = torch.zeros(256 * 64, 192, dtype=torch.bfloat16, device="cuda")
    biasd = torch.zeros(3, 64, 64, device="cuda")
    with pytest.raises(SrkUnsupported):
        ops.window_attention_bwd_fused(xn, xn[:576], None, 1.0, xn, xn[:192], biasd, 64, 64, 0)
