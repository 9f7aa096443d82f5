"""HAT on the HIP path (BASELINE cfg4): the 256-query window attention kernels against a plain torch fp32
reference of the same op, the CAB helper kernels, and the whole model against the reference's golden vectors (G13) and the
CPU oracle (oracle/hat_oracle.py, pinned by tests/test_oracle_golden.py::test_g13_*).

Tolerances as for SwinIR (tests/test_gpu_model.py): bf16 MFMA operands with fp32 accumulation / softmax / LayerNorm /
residual stream -> forward max|err| <= 1.2e-2 * max|ref|; attention kernels alone (bf16 in, bf16 out) 2e-2 * max|ref|."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import hat_oracle as HO
from oracle import swinir_oracle as O
from test_oracle_golden import hat_tiny_weights

pytestmark = pytest.mark.gpu


def _lib():
    from tpu_superresolution_amd._lib import check, lib
    return check, lib()


def _attn_reference(qkv, bias, B, H, W, ws, shift, nH, scale, overlap):
    """fp32 torch restatement on the bf16-rounded inputs: q/k/v [B, H*W, nH, 32] from the raster qkv tensor."""
    CA = nH * 32
    t = qkv.float().reshape(B, H * W, 3, nH, 32)
    q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
    qi = torch.from_numpy(O.window_token_index(H, W, ws, shift))                      # [nW, 256]
    nW = qi.shape[0]
    qw = q[:, qi.reshape(-1)].reshape(B, nW, ws * ws, nH, 32).permute(0, 1, 3, 2, 4)
    if overlap:
        ki_np, valid_np = HO.overlap_window_index(H, W, ws, ws + overlap)
        ki, valid = torch.from_numpy(ki_np), torch.from_numpy(valid_np)
        kw = (k[:, ki.reshape(-1)].reshape(B, nW, -1, nH, 32) * valid[None, :, :, None, None]).permute(0, 1, 3, 2, 4)
        vw = (v[:, ki.reshape(-1)].reshape(B, nW, -1, nH, 32) * valid[None, :, :, None, None]).permute(0, 1, 3, 2, 4)
    else:
        kw = k[:, qi.reshape(-1)].reshape(B, nW, ws * ws, nH, 32).permute(0, 1, 3, 2, 4)
        vw = v[:, qi.reshape(-1)].reshape(B, nW, ws * ws, nH, 32).permute(0, 1, 3, 2, 4)
    s = (qw @ kw.transpose(-2, -1)) * scale + bias[None, None]
    if shift and not overlap:
        s = s + torch.from_numpy(O.shift_attn_mask(H, W, ws, shift))[None, :, None]
    out = (s.softmax(-1) @ vw).permute(0, 1, 3, 2, 4).reshape(B, nW * ws * ws, CA)      # window order
    res = torch.zeros(B, H * W, CA)
    res[:, qi.reshape(-1)] = out
    return res.reshape(B * H * W, CA)


@pytest.mark.parametrize("shift,overlap,H,W", [(0, 0, 32, 48), (8, 0, 32, 48), (8, 0, 64, 32), (0, 8, 32, 48), (0, 8, 16, 16)])
def test_win256_attention_forward_vs_torch(shift, overlap, H, W):
    check, L = _lib()
    B, nH, ws = 2, 3, 16
    CA = nH * 32
    g = torch.Generator().manual_seed(H * 7 + W + shift + overlap)
    qkv = (torch.randn(B * H * W, 3 * CA, generator=g) * 0.8).to(torch.bfloat16)
    qkv.view(B * H * W, 3, nH, 32)[..., 30:] = 0                                          # head_dim 30 zero-padded to 32
    NK = (ws + overlap) ** 2 if overlap else ws * ws
    bias = torch.randn(nH, 256, NK, generator=g) * 0.5
    scale = 30 ** -0.5
    ref = _attn_reference(qkv, bias, B, H, W, ws, shift, nH, scale, overlap)
    out = torch.empty(B * H * W, CA, dtype=torch.bfloat16, device="cuda")
    q_d, b_d = qkv.cuda(), bias.cuda()
    check(L.srk_win256_attention_fwd(q_d.data_ptr(), 3 * CA, CA, b_d.data_ptr(), 0, out.data_ptr(), CA, B, H, W, ws, ws, shift, shift, nH,
                                     scale, overlap, torch.cuda.current_stream().cuda_stream))
    got = out.cpu().float()
    err = float((got - ref).abs().max())
    assert err <= 2e-2 * float(ref.abs().max()), f"max err {err:.3e} vs max|ref| {float(ref.abs().max()):.3e}"
    assert float((got.view(-1, nH, 32)[..., 30:]).abs().max()) == 0.0                      # pad channels stay zero
    # table mode: the kernel indexes the bias table itself through the closed-form relative position index -- must give exactly
    # what the dense expansion of the same table (oracle: negative rpi_oca wrapped) gives
    rows = (2 * ws + overlap - 1) ** 2
    table = torch.randn(rows, nH, generator=g) * 0.5
    dense = HO.oca_bias(table, ws, ws + overlap) if overlap else HO.sa_bias(table, ws)
    t_d, d_d = table.cuda(), dense.cuda()
    out_t, out_d = torch.empty_like(out), torch.empty_like(out)
    for src, n_rows, dst in ((t_d, rows, out_t), (d_d, 0, out_d)):
        check(L.srk_win256_attention_fwd(q_d.data_ptr(), 3 * CA, CA, src.data_ptr(), n_rows, dst.data_ptr(), CA, B, H, W, ws, ws, shift, shift,
                                         nH, scale, overlap, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out_t, out_d)


def test_channel_gate_and_cab_add_ln_vs_torch():
    check, L = _lib()
    B, HW, C, CP, S = 3, 700, 180, 192, 6
    g = torch.Generator().manual_seed(1)
    conv = torch.zeros(B * HW, CP)
    conv[:, :C] = torch.randn(B * HW, C, generator=g)
    conv = conv.to(torch.bfloat16)
    w1, b1, w2, b2 = torch.randn(S, C, generator=g) * 0.3, torch.randn(S, generator=g), torch.randn(C, S, generator=g), torch.randn(C, generator=g)
    mean = conv.float().reshape(B, HW, CP)[:, :, :C].mean(1)
    gate_ref = 0.01 * torch.sigmoid(torch.relu(mean @ w1.t() + b1) @ w2.t() + b2)
    ws_ = torch.empty(int(L.srk_channel_gate_workspace(B, HW, CP)), dtype=torch.uint8, device="cuda")
    gate = torch.empty(B, CP, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    dev = [t.cuda() for t in (conv, w1, b1, w2, b2)]
    check(L.srk_channel_gate(dev[0].data_ptr(), ws_.data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), 0.01,
                             gate.data_ptr(), B, HW, C, CP, S, st))
    assert float((gate.cpu()[:, :C] - gate_ref).abs().max()) <= 1e-6 and float(gate.cpu()[:, C:].abs().max()) == 0.0
    x = torch.zeros(B * HW, CP)
    x[:, :C] = torch.randn(B * HW, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    want = x[:, :C] + conv.float()[:, :C] * gate_ref.repeat_interleave(HW, 0)
    want_n = torch.nn.functional.layer_norm(want, (C,), gamma, beta, 1e-5)
    xd, xn = x.cuda(), torch.empty(B * HW, CP, dtype=torch.bfloat16, device="cuda")
    gd, bd = gamma.cuda(), beta.cuda()          # keep the device copies alive across the call
    check(L.srk_cab_add_ln(xd.data_ptr(), dev[0].data_ptr(), gate.data_ptr(), gd.data_ptr(), bd.data_ptr(), xn.data_ptr(),
                           B * HW, HW, C, CP, st))
    assert float((xd.cpu()[:, :C] - want).abs().max()) <= 1e-5 and float(xd.cpu()[:, C:].abs().max()) == 0.0
    assert float((xn.cpu().float()[:, :C] - want_n).abs().max()) <= 2e-2


def _build(cfg, sd):
    import tpu_superresolution_amd as T
    m = T.HAT(**cfg.kwargs())
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert list(m.state_dict().keys()) == list(sd.keys())
    return m.cuda().eval()


def test_hat_tiny_forward_vs_reference_golden():
    g, cfg, sd = hat_tiny_weights()
    m = _build(cfg, sd)
    for hw in ((32, 32), (32, 48), (20, 37)):         # exact multiple / non-square / reflect-pad + crop + dynamic mask
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = m(x.cuda()).cpu()
        ref = torch.from_numpy(g[f"y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape and y.dtype == torch.float32
        err = float((y - ref).abs().max())
        print(f"HAT tiny {hw}: max err {err:.3e} ({err / float(ref.abs().max()):.2e} of range)")
        assert err <= 1.2e-2 * float(ref.abs().max()), f"{hw}: max err {err:.3e} vs ref max {float(ref.abs().max()):.3e}"
    out_sd = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(out_sd[k].cpu(), v), k


def test_hat_cfg4_forward_probes_and_batch():
    """HAT-SRx4 (BASELINE cfg4: window 16, dim 180, 6x6 blocks): one image against the reference's probes, then the cfg4 batch
    (bs 16) against the single-image run (no cross-sample operator: the ChannelAttention pool is per sample)."""
    g = load_golden("g13_hat_cfg4_probe")
    cfg = HO.HATConfig.sr_x4()
    sd = HO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    m = _build(cfg, sd)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) and len(m.state_dict()) == int(g["n_keys"])
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
    assert tuple(y.shape) == tuple(g["shape"])
    err = np.abs(y.numpy().reshape(-1)[g["probe_index"]] - g["probe_value"]).max()
    print(f"HAT cfg4: probe max err {err:.3e}, mean {float(y.mean()):.5f} (ref {float(g['mean']):.5f})")
    assert err <= 5e-3 and abs(float(y.mean()) - float(g["mean"])) <= 2e-3      # default-scale weights: output in the image range
    xb = torch.rand(16, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    xb[5] = x[0]
    with torch.no_grad():
        yb = m(xb.cuda()).cpu()
    assert yb.shape == (16, 3, 256, 256) and torch.isfinite(yb).all()
    assert float((yb[5] - y[0]).abs().max()) <= 4e-3 * float(y.abs().max())     # different M -> different GEMM kernels / summation order


def test_hat_errors_are_loud():
    import tpu_superresolution_amd as T
    cfg = HO.HATConfig(**{**HO.HATConfig.sr_x4().__dict__, "depths": (1,), "num_heads": (6,)})
    m = T.HAT(**cfg.kwargs())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 32, 32))
    with pytest.raises(NotImplementedError, match="window_size=8"):
        T.HAT(**{**cfg.kwargs(), "window_size": 8}).cuda().eval()(torch.rand(1, 3, 32, 32, device="cuda"))


# ---- training (csrc/attn256_bwd.hip, csrc/hat_train.hip, tpu_superresolution_amd/hat_train.py) -------------------------------------------
@pytest.mark.parametrize("shift,overlap,H,W", [(0, 0, 32, 48), (8, 0, 32, 48), (8, 0, 64, 32), (0, 8, 32, 48), (0, 8, 16, 16)])
def test_win256_attention_backward_vs_autograd(shift, overlap, H, W):
    """d q / d k / d v / d table of the table-indexed 256-query window attention against autograd on the fp32 restatement (same
    bf16-rounded inputs).  The overlapping form sums the key / value gradients of a token over the key windows that hold it and
    drops the zero-padded keys; negative relative_position_index_OCA entries land on the wrapped table rows."""
    check, L = _lib()
    B, nH, ws = 2, 3, 16
    CA = nH * 32
    g = torch.Generator().manual_seed(H * 5 + W + shift + overlap)
    qkv = (torch.randn(B * H * W, 3 * CA, generator=g) * 0.8).to(torch.bfloat16)
    qkv.view(B * H * W, 3, nH, 32)[..., 30:] = 0
    dout = (torch.randn(B * H * W, CA, generator=g) * 0.5).to(torch.bfloat16)
    dout.view(B * H * W, nH, 32)[..., 30:] = 0
    rows = (2 * ws + overlap - 1) ** 2
    table = torch.randn(rows, nH, generator=g) * 0.5
    scale = 30 ** -0.5
    qr = qkv.float().clone().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    dense = HO.oca_bias(tr, ws, ws + overlap) if overlap else HO.sa_bias(tr, ws)
    _attn_reference(qr, dense, B, H, W, ws, shift, nH, scale, overlap).backward(dout.float())
    st = torch.cuda.current_stream().cuda_stream
    q_d, o_d, t_d = qkv.cuda(), dout.cuda(), table.cuda()
    dqkv = torch.zeros(B * H * W, 3 * CA, dtype=torch.bfloat16, device="cuda")
    dtab = torch.zeros(rows, nH, device="cuda")
    scratch = torch.empty(int(L.srk_win256_attention_bwd_scratch(B, H, W, nH, CA, rows, overlap)), dtype=torch.uint8, device="cuda")
    check(L.srk_win256_attention_bwd(q_d.data_ptr(), 3 * CA, CA, t_d.data_ptr(), rows, o_d.data_ptr(), CA, dqkv.data_ptr(), dtab.data_ptr(),
                                     scratch.data_ptr(), B, H, W, shift, shift, nH, scale, overlap, st))
    got, ref = dqkv.cpu().float().view(-1, 3, nH, 32), qr.grad.view(-1, 3, nH, 32)
    for i, name in enumerate("qkv"):
        err = float((got[:, i] - ref[:, i]).abs().max())
        assert err <= 2.5e-2 * float(ref[:, i].abs().max()), f"d{name}: max err {err:.3e} vs max|ref| {float(ref[:, i].abs().max()):.3e}"
    assert float(got[..., 30:].abs().max()) == 0.0
    err_t = float((dtab.cpu() - tr.grad).abs().max())
    assert err_t <= 2e-2 * max(1.0, float(tr.grad.abs().max())), f"d table: max err {err_t:.3e} vs {float(tr.grad.abs().max()):.3e}"


def test_cab_backward_vs_autograd():
    check, L = _lib()
    B, HW, C, CP, S = 3, 700, 180, 192, 6
    g = torch.Generator().manual_seed(2)
    conv = torch.zeros(B * HW, CP)
    conv[:, :C] = torch.randn(B * HW, C, generator=g)
    conv = conv.to(torch.bfloat16)
    grad = torch.zeros(B * HW, CP)
    grad[:, :C] = torch.randn(B * HW, C, generator=g)
    w1, b1, w2, b2 = torch.randn(S, C, generator=g) * 0.3, torch.randn(S, generator=g), torch.randn(C, S, generator=g), torch.randn(C, generator=g)
    cr = conv.float()[:, :C].clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    gate_ref = 0.01 * torch.sigmoid(torch.relu(cr.reshape(B, HW, C).mean(1) @ pr[0].t() + pr[1]) @ pr[2].t() + pr[3])
    (cr * gate_ref.repeat_interleave(HW, 0)).backward(grad[:, :C])
    st = torch.cuda.current_stream().cuda_stream
    dev = [t.cuda() for t in (conv, w1, b1, w2, b2, grad)]
    ws_ = torch.empty(int(L.srk_channel_gate_workspace(B, HW, CP)), dtype=torch.uint8, device="cuda")
    gate = torch.empty(B, CP, device="cuda")
    check(L.srk_channel_gate(dev[0].data_ptr(), ws_.data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), 0.01,
                             gate.data_ptr(), B, HW, C, CP, S, st))
    wsb = torch.empty(int(L.srk_cab_bwd_workspace(B, HW, CP)), dtype=torch.uint8, device="cuda")
    dw1, db1, dw2, db2 = (torch.zeros_like(t, device="cuda") for t in (w1, b1, w2, b2))
    dmean, dconv = torch.empty(B, CP, device="cuda"), torch.empty(B * HW, CP, dtype=torch.bfloat16, device="cuda")
    check(L.srk_cab_bwd(dev[0].data_ptr(), dev[5].data_ptr(), gate.data_ptr(), wsb.data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                        dev[3].data_ptr(), dev[4].data_ptr(), 0.01, dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                        dmean.data_ptr(), dconv.data_ptr(), B, HW, C, CP, S, st))
    for got, ref, name in ((dw1, pr[0].grad, "dw1"), (db1, pr[1].grad, "db1"), (dw2, pr[2].grad, "dw2"), (db2, pr[3].grad, "db2")):
        assert float((got.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())) + 1e-6, name
    err = float((dconv.cpu().float()[:, :C] - cr.grad).abs().max())
    assert err <= 1e-2 * float(cr.grad.abs().max()) and float(dconv.cpu().float()[:, C:].abs().max()) == 0.0


def test_hat_tiny_gradients_vs_reference_golden():
    """G13's training record (reference HAT, drop_path 0, L1 loss on a 2 x 3 x 32 x 32 batch): loss, the stored gradient tensors
    (OCAB of layer 1, CAB of a shifted HAB, conv_first, conv_last) and the gradient norm of every parameter."""
    import tpu_superresolution_amd as T
    g, cfg, sd = hat_tiny_weights()
    m = T.HAT(drop_path_rate=0.0, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(int(g["train.x_seed"])))
    t = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(int(g["train.target_seed"])))
    loss = torch.nn.functional.l1_loss(m(x.cuda()), t.cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(g["train.loss"])) <= 2e-3 * float(g["train.loss"])
    rels = []
    params = dict(m.named_parameters())
    for k in g.files:
        if not k.startswith("grad."):
            continue
        n, ref = k[5:], torch.from_numpy(g[k])
        p = params[n]
        assert p.grad is not None and p.grad.shape == ref.shape, n
        rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
        assert rel <= 0.1, f"{n}: relative L2 error {rel:.3e}"
        rels.append(rel)
    assert float(np.median(rels)) <= 0.04
    names, norms = [str(s) for s in g["train.grad_names"]], g["train.grad_norms"]
    assert names == [n for n, _ in m.named_parameters()]
    for n, ref in zip(names, norms):
        got = float(params[n].grad.norm())
        assert abs(got - float(ref)) <= 0.1 * float(ref) + 1e-7, f"{n}: |grad| {got:.4e} vs reference {float(ref):.4e}"
    # accumulation semantics: a second backward doubles the gradients
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    torch.nn.functional.l1_loss(m(x.cuda()), t.cuda()).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=2e-3, atol=2e-6 * float(g1[n].abs().max()) + 1e-9), n


def test_hat_train_step_with_drop_path_runs_and_learns():
    """Train mode with the reference's default drop_path_rate (0.1): DropPath factors drawn per block and sample, a few AdamW steps
    lower the loss."""
    import tpu_superresolution_amd as T
    g, cfg, sd = hat_tiny_weights()
    m = T.HAT(drop_path_rate=0.1, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3, weight_decay=0.0)
    torch.manual_seed(0)
    x = torch.rand(2, 3, 32, 32, device="cuda")
    t = torch.rand(2, 3, 128, 128, device="cuda")
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(m(x), t)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_hat_width_180_backward_fused_mlp_kernel_vs_separate_kernels():
    """At the HAT-SRx4 width with >= 64 * #CUs tokens the MLP half of every block's backward runs as ONE kernel (srk_mlp_fused_bwd: fc2 dgrad *
    GELU' -> fc1 dgrad -> norm2 backward, bf16 copy scaled by the DropPath factor): same gradients as the separate GEMMs + LayerNorm
    backward (option mlp_bwd_fused = 0), DropPath factors identical in both runs."""
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd._lib import check, lib
    cfg = HO.HATConfig(**{**HO.HATConfig.sr_x4().__dict__, "depths": (2,), "num_heads": (6,)})
    sd = HO.random_state_dict(cfg, seed=3, scale=1.0)
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 64, 64, generator=gen).cuda()          # 16 384 tokens
    t = torch.rand(4, 3, 256, 256, generator=gen).cuda()
    res = {}
    try:
        for on in (1, 0):
            check(lib().srk_set_option(b"mlp_bwd_fused", on))
            m = T.HAT(drop_path_rate=0.2, **cfg.kwargs())
            m.load_state_dict(sd, strict=True)
            m = m.cuda().train()
            torch.manual_seed(5)
            torch.cuda.manual_seed(5)
            out = m(x)
            torch.nn.functional.l1_loss(out, t).backward()
            res[on] = (out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()})
    finally:
        check(lib().srk_set_option(b"mlp_bwd_fused", 1))
    assert torch.equal(res[1][0], res[0][0])
    worst = 0.0
    for n in res[0][1]:
        rel = float((res[1][1][n] - res[0][1][n]).norm() / (res[0][1][n].norm() + 1e-12))
        worst = max(worst, rel)
        # the fused kernel keeps d xn2 in fp32 between the fc1 dgrad and the LayerNorm backward and rounds the DropPath-scaled copy once
        # (the separate path rounds d xn2 to bf16 and scales a bf16 copy): bf16-level differences, largest on the bias-table gradients
        assert rel <= 1e-2, (n, rel)
    print("worst relative gradient difference", worst)


def test_hat_width_180_train_step_vs_oracle_autograd():
    """HAT-SRx4 width (embed 180, 6 heads, window 16, CAB, OCAB; one group of two HABs) with 16 384 tokens -- where the persistent
    streaming GEMMs, the fused MLP forward / backward kernels and the OCAB's fused LayerNorm-backward epilogue run -- against autograd
    over the CPU oracle (drop_path 0): loss, output and every parameter's gradient."""
    import tpu_superresolution_amd as T
    cfg = HO.HATConfig(**{**HO.HATConfig.sr_x4().__dict__, "depths": (2,), "num_heads": (6,), "upscale": 2})
    sd = HO.random_state_dict(cfg, seed=8, scale=1.0)
    m = T.HAT(drop_path_rate=0.0, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    gen = torch.Generator().manual_seed(4)
    x, t = torch.rand(4, 3, 64, 64, generator=gen), torch.rand(4, 3, 128, 128, generator=gen)
    y = m(x.cuda())
    loss = torch.nn.functional.l1_loss(y, t.cuda())
    loss.backward()
    leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and k in dict(m.named_parameters()) else v) for k, v in sd.items()}
    yo = HO.hat_forward(leaf, cfg, x)
    lo = (yo - t).abs().mean()
    names = [n for n, _ in m.named_parameters()]
    grads = dict(zip(names, torch.autograd.grad(lo, [leaf[n] for n in names], allow_unused=True)))
    assert float((y.detach().cpu() - yo.detach()).abs().max()) <= 2e-2 * float(yo.abs().max())
    assert abs(float(loss.detach()) - float(lo.detach())) <= 5e-3 * float(lo.detach())
    biggest = max(float(g.norm()) for g in grads.values() if g is not None)
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        w = grads[n] if grads[n] is not None else torch.zeros_like(p.detach().cpu())
        e = float((p.grad.cpu().float() - w).norm()) / max(float(w.norm()), 2e-3 * biggest)
        if e > worst[1]:
            worst = (n, e)
    print(f"worst gradient error {worst[1]:.3e} at {worst[0]}")
    assert worst[1] <= 0.1, worst


def test_hat_graphed_train_step_matches_eager_steps():
    """training.GraphedTrainStep on HAT (incl. the overlapping cross-attention backward, whose fp32 accumulation image must be re-zeroed by
    every replay: a hipMemsetAsync node did not do that under capture on this ROCm build, the library now zeroes with its own kernel):
    losses of four replayed steps on changing batches against the same steps launched eagerly (drop_path 0)."""
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd.training import GraphedTrainStep, l1_loss_checked
    g, cfg, sd = hat_tiny_weights()
    gen = torch.Generator().manual_seed(9)
    batches = [(torch.rand(2, 3, 32, 32, generator=gen).cuda(), torch.rand(2, 3, 128, 128, generator=gen).cuda()) for _ in range(4)]

    def model():
        m = T.HAT(drop_path_rate=0.0, **cfg.kwargs())
        m.load_state_dict(sd, strict=True)
        return m.cuda().train()
    ma, mb = model(), model()
    oa = torch.optim.AdamW(ma.parameters(), lr=1e-4, weight_decay=0.0)
    ob = torch.optim.AdamW(mb.parameters(), lr=1e-4, weight_decay=0.0, capturable=True)
    gs = GraphedTrainStep(mb, ob, max_grad_norm=1.0, warmup=1)

    def eager(x, t):
        oa.zero_grad(set_to_none=True)
        loss, _ = l1_loss_checked(ma(x), t)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(ma.parameters(), 1.0)
        oa.step()
        return float(loss.detach()), float(gn)
    eager(*batches[0])                   # the graphed stepper warms up with one eager step on its first batch
    la, lb = [], []
    for x, t in batches:
        la.append(eager(x, t)[0])
        lg, bad = gs(x, t)
        lb.append(float(lg))
        assert int(bad) == 0
    print("eager", la, "graphed", lb)
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(la, lb))
    # the gradients the last replay left behind are those of an eager backward at the same weights (not sums over replays)
    x, t = batches[-1]
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.isfinite(pb.grad).all(), n
    ga = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in ma.parameters())))
    gb = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in mb.parameters())))
    assert abs(ga - gb) <= 0.05 * ga, (ga, gb)          # both were clipped to the same norm from nearly the same raw gradients


def test_graphed_train_step_draws_fresh_drop_path_factors_every_step():
    """With drop_path > 0 the graphed step reads its DropPath factors from a static buffer that is re-drawn (eagerly) before every replay:
    the factors differ from step to step, some are zero, and the steps stay finite and lower the loss."""
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd.training import GraphedTrainStep
    g, cfg, sd = hat_tiny_weights()
    m = T.HAT(drop_path_rate=0.4, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3, weight_decay=0.0, capturable=True)
    gs = GraphedTrainStep(m, opt, max_grad_norm=1.0, warmup=1)
    torch.manual_seed(3)
    x, t = torch.rand(2, 3, 32, 32, device="cuda"), torch.rand(2, 3, 128, 128, device="cuda")
    seen, losses = [], []
    for _ in range(6):
        loss, bad = gs(x, t)
        seen.append(gs.drop.clone())
        losses.append(float(loss))
        assert int(bad) == 0
    assert gs.drop is not None and m._drop_override is gs.drop
    assert any(not torch.equal(seen[i], seen[i + 1]) for i in range(5)) and any(float(s.min()) == 0.0 for s in seen)
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0], losses
    gs.close()
    assert m._drop_override is None
