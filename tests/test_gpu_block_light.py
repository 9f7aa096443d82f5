"""srk_swin_block_fwd (csrc/block_light.hip: SwinTransformerBlock.forward, network_swinir.py:239-279, as one kernel at the
SwinIR-light width) called through the C ABI against the CPU oracle's swin_block on the same seeded inputs.

Tolerance: the kernel rounds where the layer-per-launch bf16 path rounds (LayerNorm outputs, q/k/v, attention probabilities,
attention output, hidden activations; fp32 residual stream), so one block deviates from the fp32 oracle by bf16 noise only:
max |err| <= 2e-2 * max |branch update|, measured ~6e-3.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import swinir_oracle as O

pytestmark = pytest.mark.gpu


def _block_weights(Cdim, nH, hid, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    pre = "b."
    return pre, {
        pre + "norm1.weight": 1.0 + 0.2 * r(Cdim), pre + "norm1.bias": 0.1 * r(Cdim),
        pre + "attn.relative_position_bias_table": 0.5 * r(225, nH),
        pre + "attn.qkv.weight": r(3 * Cdim, Cdim, sc=Cdim ** -0.5), pre + "attn.qkv.bias": 0.1 * r(3 * Cdim),
        pre + "attn.proj.weight": r(Cdim, Cdim, sc=Cdim ** -0.5), pre + "attn.proj.bias": 0.1 * r(Cdim),
        pre + "norm2.weight": 1.0 + 0.2 * r(Cdim), pre + "norm2.bias": 0.1 * r(Cdim),
        pre + "mlp.fc1.weight": r(hid, Cdim, sc=Cdim ** -0.5), pre + "mlp.fc1.bias": 0.1 * r(hid),
        pre + "mlp.fc2.weight": r(Cdim, hid, sc=hid ** -0.5), pre + "mlp.fc2.bias": 0.1 * r(Cdim),
    }


@pytest.mark.parametrize("shift", [0, 4])
@pytest.mark.parametrize("dims", [(60, 6, 120), (48, 6, 96)])
def test_whole_block_kernel_matches_oracle_block(shift, dims):
    from tpu_superresolution_amd._lib import check, lib
    from tpu_superresolution_amd.hat_arch import _head_map, _pack_linear, _pack_vec
    Cdim, nH, hid = dims
    dh = Cdim // nH
    B, H, W = 3, 24, 40                                   # 15 windows per image; masked border windows when shifted
    pre, sd = _block_weights(Cdim, nH, hid, seed=7 + shift)
    x = torch.randn(B, H * W, Cdim, generator=torch.Generator().manual_seed(1))
    ref = O.swin_block(x, (H, W), sd, pre, nH, 8, shift)

    dev = torch.device("cuda")
    hm = _head_map(nH, dh, dev)
    qmap = torch.cat([w * 192 + hm for w in range(3)])     # qkv row which * C + h * dh + d -> which * 192 + h * 32 + d
    cu = lambda t: t.to(dev)
    wqkv = _pack_linear(cu(sd[pre + "attn.qkv.weight"]), 576, 64, row_map=qmap)
    bqkv = _pack_vec(cu(sd[pre + "attn.qkv.bias"]), 576, row_map=qmap)
    wproj = _pack_linear(cu(sd[pre + "attn.proj.weight"]), 64, 192, col_map=hm)
    bproj = _pack_vec(cu(sd[pre + "attn.proj.bias"]), 64)
    w1 = _pack_linear(cu(sd[pre + "mlp.fc1.weight"]), 128, 64)
    b1 = _pack_vec(cu(sd[pre + "mlp.fc1.bias"]), 128)
    w2 = _pack_linear(cu(sd[pre + "mlp.fc2.weight"]), 64, 128)
    b2 = _pack_vec(cu(sd[pre + "mlp.fc2.bias"]), 64)
    dense = cu(O.dense_rel_pos_bias(sd[pre + "attn.relative_position_bias_table"], 8).contiguous().float())
    xp = torch.zeros(B * H * W, 64, device=dev)
    xp[:, :Cdim] = cu(x).reshape(-1, Cdim)
    y = torch.empty_like(xp)
    yb = torch.empty(B * H * W, 64, dtype=torch.bfloat16, device=dev)
    n1w, n1b, n2w, n2b = (cu(sd[pre + k]).float().contiguous() for k in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"))
    st = torch.cuda.current_stream().cuda_stream
    args = (xp.data_ptr(), y.data_ptr(), yb.data_ptr(), n1w.data_ptr(), n1b.data_ptr(), n2w.data_ptr(), n2b.data_ptr(), wqkv.data_ptr(),
            bqkv.data_ptr(), wproj.data_ptr(), bproj.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), dense.data_ptr(),
            float(dh ** -0.5), Cdim, nH, dh, hid, B, H, W, shift, st)
    check(lib().srk_swin_block_fwd(*args))
    torch.cuda.synchronize()
    got = y[:, :Cdim].reshape(B, H * W, Cdim).cpu()
    assert torch.isfinite(y).all() and float(y[:, Cdim:].abs().max()) == 0.0       # pad channels stay zero
    upd = float((ref - x).abs().max())
    err = float((got - ref).abs().max())
    print(f"C={Cdim} shift={shift}: max err {err:.3e} (branch update max {upd:.3e})")
    assert err <= 2e-2 * upd
    assert torch.equal(yb.float(), y.to(torch.bfloat16).float())
    # in place (y aliases x): windows are disjoint and every row is read before it is written
    check(lib().srk_swin_block_fwd(xp.data_ptr(), xp.data_ptr(), None, *args[3:]))
    torch.cuda.synchronize()
    assert torch.equal(xp, y)
    # another head layout of the same width (3 heads x 20) has no whole-block kernel, and the entry says so instead of computing
    bad = list(args)
    bad[18], bad[19] = 3, 20
    if Cdim == 60:
        assert lib().srk_swin_block_fwd(*bad) != 0
        assert b"light width" in lib().srk_last_error()
