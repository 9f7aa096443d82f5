"""Parity of the individual HIP kernels (through the C ABI) against the CPU oracle / plain fp32 math.

Tolerances: index ops bit-exact; bf16-operand GEMM-like kernels are compared with an fp32 reference
computed from the SAME bf16-rounded operands, so the only differences are fp32 accumulation order
and the final bf16 rounding of the output (rel 2^-8).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import swinir_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tpu_superresolution_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


def bf(t):
    return t.to(torch.bfloat16)


def close_bf16(got, ref, atol, rtol=2 ** -7):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f"max err {err.max():.4e} at {int(bad.sum())} / {bad.numel()} elements (ref max {ref.abs().max():.3e})"


# ---------------------------------------------------------------------------------------------------
def test_trread_contract(ops):
    tile = torch.arange(64 * 16, dtype=torch.int16).reshape(64, 16)
    out = ops.probe_trread(dev(tile)).cpu()
    lanes = torch.arange(64)
    exp = torch.stack([tile[8 * (lanes >> 4) + j, lanes & 15] for j in range(8)], dim=1)
    assert torch.equal(out, exp), f"ds_read_b64_tr_b16 contract differs:\n{out[:4]}\nvs\n{exp[:4]}"


def test_window_partition_reverse_roll_bit_exact(ops):
    g = load_golden("g1_g2_index_maps")
    x = torch.from_numpy(g["x"])
    wp = ops.window_partition(dev(x), 8)
    assert torch.equal(wp.cpu(), torch.from_numpy(g["partition"]))
    assert torch.equal(ops.window_reverse(wp, 8, 16, 24).cpu(), x)
    assert torch.equal(ops.roll2d(dev(x), (-4, -4)).cpu(), torch.from_numpy(g["roll_m4"]))
    assert torch.equal(ops.roll2d(dev(x), (4, 4)).cpu(), torch.from_numpy(g["roll_p4"]))
    assert torch.equal(ops.window_partition(ops.roll2d(dev(x), (-4, -4)), 8).cpu(), torch.from_numpy(g["roll_m4_partition"]))
    # other element sizes / ragged channel counts / window 7
    for dtype, C in ((torch.int16, 5), (torch.int64, 1), (torch.float32, 180)):
        y = torch.arange(3 * 14 * 21 * C).reshape(3, 14, 21, C).to(dtype)
        got = ops.window_partition(dev(y), 7).cpu()
        assert torch.equal(got, torch.from_numpy(O.np_window_partition(y.numpy(), 7)))
        assert torch.equal(ops.window_reverse(dev(got), 7, 14, 21).cpu(), y)


def test_window_partition_cfg3_size_sha1(ops):
    import hashlib
    g = load_golden("g1_g2_index_maps")
    big = torch.arange(32 * 64 * 64 * 180, dtype=torch.int32).reshape(32, 64, 64, 180)
    out = ops.window_partition(dev(big), 8).cpu().numpy()
    assert hashlib.sha1(np.ascontiguousarray(out).tobytes()).hexdigest() == str(g["cfg3_partition_sha1"])
    out2 = ops.window_partition(ops.roll2d(dev(big), (-4, -4)), 8).cpu().numpy()
    assert hashlib.sha1(np.ascontiguousarray(out2).tobytes()).hexdigest() == str(g["cfg3_roll_partition_sha1"])


def test_pixel_shuffle_mask_rpi_bit_exact(ops):
    g = load_golden("g7_upsample")
    x = torch.from_numpy(g["ps.x"])
    assert torch.equal(ops.pixel_shuffle(dev(x), 2).cpu(), torch.from_numpy(g["ps.r2"]))
    assert torch.equal(ops.pixel_shuffle(dev(x), 3).cpu(), torch.from_numpy(g["ps.r3"]))
    gm = load_golden("g4_masks")
    for hw in ((64, 64), (48, 48), (16, 24), (24, 40)):
        m = ops.shift_mask(hw[0], hw[1], 8, 4).cpu()
        assert set(torch.unique(m).tolist()) <= {0.0, -100.0}
        assert torch.equal((m != 0).to(torch.uint8), torch.from_numpy(gm[f"mask_{hw[0]}x{hw[1]}"]))
    gr = load_golden("g3_rpi")
    for ws in (7, 8, 16):
        assert torch.equal(ops.relative_position_index(ws).cpu(), torch.from_numpy(gr[f"rpi_ws{ws}"]))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,CP,geom", [(180, 192, None), (180, 192, (16, 24, 4)), (60, 64, (16, 16, 0)), (24, 64, (8, 16, 4))])
def test_layernorm_fwd(ops, C, CP, geom):
    torch.manual_seed(1)
    B, H, W = (2, 16, 24) if geom is None else (2, geom[0], geom[1])
    rows = B * H * W
    x = torch.zeros(rows, CP)
    x[:, :C] = torch.randn(rows, C) * 2 + 0.5
    gamma, beta = torch.randn(C) * 0.3 + 1, torch.randn(C) * 0.2
    yb, yf, mean, rstd = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), C, geom=geom, out_f32=True)
    ref = torch.nn.functional.layer_norm(x[:, :C], (C,), gamma, beta, 1e-5)
    if geom is not None:
        idx = torch.from_numpy(O.window_token_index(H, W, 8, geom[2])).reshape(-1)
        ref = ref.view(B, H * W, C)[:, idx].reshape(rows, C)
    assert (yf.cpu()[:, :C] - ref).abs().max() < 2e-5
    assert (yf.cpu()[:, C:] == 0).all() and (yb.cpu()[:, C:] == 0).all()
    close_bf16(yb[:, :C], ref, 1e-3)


@pytest.mark.parametrize("M,N,K", [(256, 192, 192), (192, 576, 192), (384, 384, 192), (128, 64, 384), (200, 128, 576)])
def test_linear_bf16(ops, M, N, K):
    torch.manual_seed(2)
    a, w, b = bf(torch.randn(M, K)), bf(torch.randn(N, K) * 0.1), torch.randn(N)
    y = ops.linear_bf16(dev(a), dev(w), dev(b))
    ref = a.float() @ w.float().t() + b
    close_bf16(y, ref, 2e-3)


def test_linear_a_equals_identity_asymmetric(ops):
    """A = I with an asymmetric W catches a swapped (row, col) fragment map (guide rule: A=I check)."""
    K = N = 192
    a = torch.zeros(256, K)
    a[:K] = torch.eye(K)
    w = (torch.arange(N * K).reshape(N, K) % 251).float() / 64.0
    y = ops.linear_bf16(dev(bf(a)), dev(bf(w)), None)
    assert torch.equal(y.cpu()[:K].float(), bf(w).float().t())


@pytest.mark.parametrize("M,N,K", [(1024, 192, 192), (4096, 576, 192), (640, 384, 192), (512, 192, 384), (700, 64, 128), (512, 128, 64)])
def test_linear_wgrad(ops, M, N, K):
    torch.manual_seed(3)
    y, x = bf(torch.randn(M, N)), bf(torch.randn(M, K))
    dw, db = ops.linear_wgrad_bf16(dev(y), dev(x))
    ref = y.float().t() @ x.float()
    assert (dw.cpu() - ref).abs().max() < 1e-3 * max(1.0, float(ref.abs().max()))
    assert (db.cpu() - y.float().sum(0)).abs().max() < 1e-3 * max(1.0, float(y.float().sum(0).abs().max()))


def pack_conv_weight(w, NP, CinP):
    """[Cout, Cin, 3, 3] fp32 -> bf16 [NP, 9*CinP] tap-major (the kernel layout)."""
    Cout, Cin = w.shape[:2]
    out = torch.zeros(NP, 9, CinP)
    out[:Cout, :, :Cin] = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    return bf(out.reshape(NP, 9 * CinP))


@pytest.mark.parametrize("B,H,W,Cin,CinP,Cout,NP", [(2, 16, 24, 60, 64, 60, 64), (1, 8, 8, 180, 192, 180, 192), (3, 13, 9, 64, 64, 256, 256)])
def test_conv3x3(ops, B, H, W, Cin, CinP, Cout, NP):
    torch.manual_seed(4)
    x = torch.zeros(B, H, W, CinP)
    x[..., :Cin] = torch.randn(B, H, W, Cin)
    w, b = torch.randn(Cout, Cin, 3, 3) * 0.05, torch.randn(Cout)
    bp = torch.zeros(NP)
    bp[:Cout] = b
    y = ops.conv3x3_bf16(dev(bf(x)), dev(pack_conv_weight(w, NP, CinP)), dev(bp))
    ref = torch.nn.functional.conv2d(bf(x[..., :Cin]).float().permute(0, 3, 1, 2), bf(w).float(), b, padding=1).permute(0, 2, 3, 1)
    close_bf16(y[..., :Cout], ref, 3e-3)
    assert (y.cpu()[..., Cout:] == 0).all()


@pytest.mark.parametrize("B,H,W,CinP,N", [(2, 16, 16, 64, 64), (1, 24, 16, 192, 192), (2, 8, 8, 64, 256),
                                          (2, 8, 64, 64, 128), (1, 5, 128, 192, 192), (3, 3, 64, 128, 64)])   # W % 64 == 0: all-taps kernel
def test_conv3x3_wgrad(ops, B, H, W, CinP, N):
    torch.manual_seed(5)
    x, dy = bf(torch.randn(B, H, W, CinP)), bf(torch.randn(B, H, W, N) * 0.1)
    dw, db = ops.conv3x3_wgrad_bf16(dev(dy), dev(x))
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    wt = torch.zeros(N, CinP, 3, 3, requires_grad=True)
    out = torch.nn.functional.conv2d(xt, wt, None, padding=1)
    out.backward(dy.float().permute(0, 3, 1, 2))
    ref = wt.grad.permute(0, 2, 3, 1).reshape(N, 9 * CinP)
    assert (dw.cpu() - ref).abs().max() < 2e-3 * max(1.0, float(ref.abs().max()))
    assert (db.cpu() - dy.float().sum((0, 1, 2))).abs().max() < 1e-2


# ---------------------------------------------------------------------------------------------------
def attention_inputs(B, H, W, nH, d, seed):
    g = torch.Generator().manual_seed(seed)
    B_ = B * (H // 8) * (W // 8)
    q = torch.randn(B_, nH, 64, d, generator=g) * 0.7
    k = torch.randn(B_, nH, 64, d, generator=g) * 0.7
    v = torch.randn(B_, nH, 64, d, generator=g)
    table = torch.randn(225, nH, generator=g) * 0.5
    return bf(q).float(), bf(k).float(), bf(v).float(), table


def pad_qkv(q, k, v):
    B_, nH, N, d = q.shape
    out = torch.zeros(3, B_, nH, 64, 32)
    out[0, ..., :d], out[1, ..., :d], out[2, ..., :d] = q, k, v
    return bf(out)


def attention_ref(q, k, v, table, H, W, shift):
    """fp32 restatement of network_swinir.py:125-142 on already-scaled q."""
    B_, nH = q.shape[:2]
    s = q @ k.transpose(-1, -2) + O.dense_rel_pos_bias(table, 8)[None]
    if shift:
        mask = torch.from_numpy(O.shift_attn_mask(H, W, 8, shift))
        nW = mask.shape[0]
        s = (s.view(B_ // nW, nW, nH, 64, 64) + mask[None, :, None]).view(B_, nH, 64, 64)
    p = torch.softmax(s, -1)
    return p @ v, p


@pytest.mark.parametrize("B,H,W,nH,d,shift", [(2, 16, 16, 2, 12, 0), (2, 16, 24, 6, 30, 4), (1, 8, 8, 6, 10, 4), (32, 16, 16, 3, 32, 4)])
def test_window_attention_fwd(ops, B, H, W, nH, d, shift):
    q, k, v, table = attention_inputs(B, H, W, nH, d, 6)
    biasd = ops.rel_pos_bias_expand(dev(table))
    assert torch.equal(biasd.cpu(), O.dense_rel_pos_bias(table, 8))
    out = ops.window_attention_fwd(dev(pad_qkv(q, k, v)), biasd, H, W, shift)
    ref, _ = attention_ref(q, k, v, table, H, W, shift)          # [B_, nH, 64, d]
    got = out.float().cpu().view(-1, 64, nH, 32).permute(0, 2, 1, 3)
    close_bf16(got[..., :d], ref, 6e-3)
    assert (got[..., d:] == 0).all()


@pytest.mark.parametrize("B,H,W,nH,d,shift", [(2, 16, 16, 2, 12, 0), (2, 16, 24, 6, 30, 4), (40, 16, 16, 3, 32, 4)])
def test_window_attention_bwd(ops, B, H, W, nH, d, shift):
    q, k, v, table = attention_inputs(B, H, W, nH, d, 7)
    B_ = q.shape[0]
    g = torch.Generator().manual_seed(70)
    do = bf(torch.randn(B_, nH, 64, d, generator=g)).float()
    scale = d ** -0.5
    # oracle gradients w.r.t. unscaled q (q_scaled = q_raw * scale), k, v, table
    qr = (q / scale).requires_grad_(True)
    kr, vr, tr = k.clone().requires_grad_(True), v.clone().requires_grad_(True), table.clone().requires_grad_(True)
    o, _ = attention_ref(qr * scale, kr, vr, tr, H, W, shift)
    o.backward(do)
    dop = torch.zeros(B_, 64, nH, 32)
    dop[..., :d] = do.permute(0, 2, 1, 3)
    dqkv, dtab = ops.window_attention_bwd(dev(pad_qkv(q, k, v)), ops.rel_pos_bias_expand(dev(table)), dev(bf(dop.reshape(B_ * 64, nH * 32))),
                                          scale, H, W, shift)
    got = dqkv.float().cpu().view(B_, 64, 3, nH, 32).permute(2, 0, 3, 1, 4)     # [3, B_, nH, 64, 32]
    for i, ref in enumerate((qr.grad, kr.grad, vr.grad)):
        close_bf16(got[i][..., :d], ref, 2e-2 * float(ref.abs().max()), rtol=2e-2)
        assert (got[i][..., d:] == 0).all()
    assert (dtab.cpu() - tr.grad).abs().max() < 2e-2 * max(1.0, float(tr.grad.abs().max()))


def test_l1_loss(ops):
    torch.manual_seed(8)
    p, t = torch.rand(2, 3, 40, 40), torch.rand(2, 3, 40, 40)
    loss, d, bad = ops.l1_loss_fwd_bwd(dev(p), dev(t))
    pr = p.clone().requires_grad_(True)
    ref = O.l1_loss(pr, t)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-6 and int(bad) == 0
    assert (d.cpu() - pr.grad).abs().max() < 1e-9
    p[0, 0, 0, 0] = float("nan")
    p[1, 2, 3, 4] = float("inf")
    _, _, bad = ops.l1_loss_fwd_bwd(dev(p), dev(t), want_grad=False)
    assert int(bad) == 2
