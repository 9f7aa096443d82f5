"""DAT on the HIP path (BASELINE cfg5): the DAT-specific kernels against plain torch fp32 references of the
same ops, and the whole model (eval mode) against the reference's golden vectors (G14) and the CPU oracle
(oracle/dat_oracle.py, pinned by tests/test_oracle_golden.py::test_g14_*).  Tolerances as for SwinIR / HAT."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import dat_oracle as DO
from test_oracle_golden import dat_tiny_weights

pytestmark = pytest.mark.gpu


def _lib():
    from tpu_superresolution_amd._lib import check, lib
    return check, lib()


def _st():
    return torch.cuda.current_stream().cuda_stream


def test_dwconv_rowln_gates_vs_torch():
    check, L = _lib()
    B, H, W, C = 2, 11, 21, 40                       # 5 groups of 8 channels; H, W not multiples of the 8 x 16 tile (edge predication)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, 64, generator=g).to(torch.bfloat16)
    w = torch.randn(C, 9, generator=g) * 0.3
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    mul = torch.randn(B * H * W, 48, generator=g).to(torch.bfloat16)
    xi = x.float()[..., 8:8 + C].permute(0, 3, 1, 2)                                   # a column slice starting at 8
    conv = F.conv2d(xi, w.reshape(C, 1, 3, 3), padding=1, groups=C) * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1)
    want = (F.gelu(conv).permute(0, 2, 3, 1).reshape(-1, C) * mul.float()[:, :C])
    xd, md = x.cuda(), mul.cuda()
    out = torch.zeros(B * H * W, 48, dtype=torch.bfloat16, device="cuda")
    wd, sd_, hd = w.cuda(), sc.cuda(), sh.cuda()
    check(L.srk_dwconv3x3(xd.data_ptr() + 8 * 2, 64, wd.data_ptr(), sd_.data_ptr(), hd.data_ptr(), md.data_ptr(), 48, out.data_ptr(), 48, B, H, W, C // 8,
                          1, _st()))
    assert float((out.cpu().float()[:, :C] - want).abs().max()) <= 2e-2 * float(want.abs().max())
    # row LayerNorm of a bf16 slice
    rows, Cn = 300, 360
    h = torch.randn(rows, 768, generator=g).to(torch.bfloat16)
    gm, bt = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1
    want = F.layer_norm(h.float()[:, 384:384 + Cn], (Cn,), gm, bt, 1e-5)
    hdv, gd, bd = h.cuda(), gm.cuda(), bt.cuda()
    o = torch.empty(rows, 384, dtype=torch.bfloat16, device="cuda")
    check(L.srk_rowln_bf16(hdv.data_ptr() + 384 * 2, 768, gd.data_ptr(), bd.data_ptr(), o.data_ptr(), 384, rows, Cn, 384, _st()))
    assert float((o.cpu().float()[:, :Cn] - want).abs().max()) <= 2e-2 and float(o.cpu().float()[:, Cn:].abs().max()) == 0.0
    # spatial gate + combine
    T, CP, S = 4 * 50, 192, 11
    a = torch.randn(T, CP, generator=g).to(torch.bfloat16)
    bb = torch.randn(T, CP, generator=g).to(torch.bfloat16)
    W0, b0, w3, b3 = torch.randn(S, CP, generator=g) * 0.1, torch.randn(S, generator=g), torch.randn(S, generator=g), 0.3
    tg_ref = torch.sigmoid(F.gelu(a.float() @ W0.t() + b0) @ w3 + b3)
    ad, bdv, W0d, b0d, w3d = a.cuda(), bb.cuda(), W0.cuda(), b0.cuda(), w3.cuda()
    tg = torch.empty(T, device="cuda")
    check(L.srk_spatial_gate(ad.data_ptr(), CP, W0d.data_ptr(), b0d.data_ptr(), w3d.data_ptr(), b3, S, tg.data_ptr(), T, CP, _st()))
    assert float((tg.cpu() - tg_ref).abs().max()) <= 2e-5
    cg = torch.rand(4, CP, generator=g)
    cgd = cg.cuda()
    out = torch.empty(T, CP, dtype=torch.bfloat16, device="cuda")
    for mode in (0, 1):
        check(L.srk_dual_gate_combine(ad.data_ptr(), bdv.data_ptr(), cgd.data_ptr(), tg.data_ptr(), out.data_ptr(), T, 50, CP, mode, _st()))
        cgr, tgr = cg.repeat_interleave(50, 0), tg.cpu()[:, None]
        want = a.float() * tgr + bb.float() * cgr if mode else a.float() * cgr + bb.float() * tgr
        assert float((out.cpu().float() - want).abs().max()) <= 2e-2 * float(want.abs().max())


@pytest.mark.parametrize("N", [256, 1000])
def test_channel_attention_vs_torch(N):
    check, L = _lib()
    B, nH, d = 2, 6, 30
    CA = nH * 32
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn(B * N, 3 * CA, generator=g)).to(torch.bfloat16)
    qkv.view(B * N, 3, nH, 32)[..., d:] = 0
    temp = torch.rand(nH, generator=g) + 0.5
    t = qkv.float().reshape(B, N, 3, nH, 32)[..., :d].permute(2, 0, 3, 4, 1)           # [3, B, h, d, N]
    attn = (F.normalize(t[0], dim=-1) @ F.normalize(t[1], dim=-1).transpose(-2, -1)) * temp.view(1, nH, 1, 1)
    want = (attn.softmax(-1) @ t[2]).permute(0, 3, 1, 2)                                # [B, N, h, d]
    qd, td = qkv.cuda(), temp.cuda()
    ws = torch.empty(int(L.srk_channel_attention_workspace(B, N, nH)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B * N, CA, dtype=torch.bfloat16, device="cuda")
    check(L.srk_channel_attention_fwd(qd.data_ptr(), 3 * CA, CA, td.data_ptr(), ws.data_ptr(), out.data_ptr(), CA, B, N, nH, d, _st()))
    got = out.cpu().float().reshape(B, N, nH, 32)
    assert float((got[..., :d] - want).abs().max()) <= 2e-2 * float(want.abs().max())
    assert float(got[..., d:].abs().max()) == 0.0


def _build(cfg, sd):
    import tpu_superresolution_amd as T
    m = T.DAT(**cfg.kwargs())
    assert list(m.state_dict().keys()) == list(sd.keys())
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.cuda().eval()


def test_dat_tiny_forward_vs_reference_golden():
    g, cfg, sd = dat_tiny_weights()
    m = _build(cfg, sd)
    for hw in ((32, 32), (32, 64)):           # the stored masks' size / another size (masks evaluated for the actual map)
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = m(x.cuda()).cpu()
        ref = torch.from_numpy(g[f"y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape and y.dtype == torch.float32
        err = float((y - ref).abs().max())
        print(f"DAT tiny {hw}: max err {err:.3e} ({err / float(ref.abs().max()):.2e} of range)")
        assert err <= 1.2e-2 * float(ref.abs().max()), f"{hw}: max err {err:.3e} vs ref max {float(ref.abs().max()):.3e}"
    for k, v in sd.items():
        assert torch.equal(m.state_dict()[k].cpu(), v), k


def test_dat_cfg5_forward_probes_and_batch():
    """DAT x4 (BASELINE cfg5: split 8x32, dim 180, 6x6 blocks, expansion 4): one image against the reference's probes, then bs 16."""
    g = load_golden("g14_dat_cfg5_probe")
    cfg = DO.DATConfig.sr_x4()
    sd = DO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    m = _build(cfg, sd)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) and len(m.state_dict()) == int(g["n_keys"])
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
    assert tuple(y.shape) == tuple(g["shape"])
    err = np.abs(y.numpy().reshape(-1)[g["probe_index"]] - g["probe_value"]).max()
    print(f"DAT cfg5: probe max err {err:.3e}, mean {float(y.mean()):.5f} (ref {float(g['mean']):.5f})")
    assert err <= 5e-3 and abs(float(y.mean()) - float(g["mean"])) <= 2e-3
    xb = torch.rand(16, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    xb[7] = x[0]
    with torch.no_grad():
        yb = m(xb.cuda()).cpu()
    assert yb.shape == (16, 3, 256, 256) and torch.isfinite(yb).all()
    assert float((yb[7] - y[0]).abs().max()) <= 4e-3 * float(y.abs().max())


def test_dat_errors_are_loud():
    import tpu_superresolution_amd as T
    cfg = DO.DATConfig(**{**DO.DATConfig.sr_x4().__dict__, "depth": (2,), "num_heads": (6,)})
    m = T.DAT(**cfg.kwargs())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(torch.rand(1, 3, 64, 64))
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):     # torch's BatchNorm error, batch of one
        m.cuda().train()(torch.rand(1, 3, 64, 64, device="cuda"))
    with pytest.raises(NotImplementedError, match="split_size"):
        T.DAT(**{**cfg.kwargs(), "split_size": [8, 8]}).cuda().eval()(torch.rand(1, 3, 64, 64, device="cuda"))


@pytest.mark.parametrize("tag", ["s832", "s816"])
def test_dat_padded_frames_and_split_8x16_vs_reference_golden(tag):
    """G14b: inputs that are not multiples of the larger split (the reference zero-pads q / k / v, dat_arch.py:376-384: the padding is
    folded into the attention kernel's load addresses, shift and mask run on the padded frame) and split_size [8, 16] with
    expansion_factor 2 -- the configuration the reference's own __main__ builds (:862-883) -- whose windows hold 128 tokens."""
    from test_oracle_golden import dat_g14b_weights
    g, cfg, sd = dat_g14b_weights(tag)
    m = _build(cfg, sd)
    sizes = ((24, 40), (48, 64), (32, 32)) if tag == "s832" else ((32, 32), (24, 40), (40, 16))
    for hw in sizes:
        x = torch.from_numpy(g[f"{tag}.x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = m(x.cuda()).cpu()
        ref = torch.from_numpy(g[f"{tag}.y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape and y.dtype == torch.float32
        err = float((y - ref).abs().max())
        print(f"DAT {tag} {hw}: max err {err:.3e} ({err / float(ref.abs().max()):.2e} of range)")
        assert err <= 1.2e-2 * float(ref.abs().max()), f"{tag} {hw}: max err {err:.3e} vs ref max {float(ref.abs().max()):.3e}"


def test_dat_cfg5_width_at_a_padded_size_vs_oracle():
    """BASELINE cfg5's width (dim 180, split 8x32, expansion 4, two groups) on a 48x64 input: 48 is padded to 64 inside the spatial
    attention.  Checked against the CPU oracle (pinned on padded frames by G14b)."""
    cfg = DO.DATConfig(**{**DO.DATConfig.sr_x4().__dict__, "depth": (3, 2), "num_heads": (6, 6)})
    sd = DO.random_state_dict(cfg, seed=5, scale=1.0)
    m = _build(cfg, sd)
    x = torch.rand(2, 3, 48, 64, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
        ref = DO.dat_forward(sd, cfg, x)
    err = float((y - ref).abs().max())
    assert y.shape == ref.shape and err <= 5e-3 * max(1.0, float(ref.abs().max())), err


# ---- training (csrc/dat_train.hip, csrc/attn_rect_bwd.hip, dat_train.py) -------------------------------------------------------------------
def _bf(t):
    return t.to(torch.bfloat16)


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


def test_dat_train_token_kernels_vs_torch():
    """chan_stats, affine_act, dgelu_affine, lincomb2, mul_bwd, dwconv3x3_wgrad, dual_gate_bwd, rowln_bwd against fp32 torch (autograd
    where it applies) on the same bf16 inputs."""
    check, L = _lib()
    g = torch.Generator().manual_seed(1)
    B, H, W, C = 2, 12, 20, 40
    HW, T = H * W, B * H * W
    ld = 64
    p, q = _bf(torch.randn(T, ld, generator=g)), _bf(torch.randn(T, ld, generator=g))
    pd, qd = p.cuda(), q.cuda()
    nck = int(L.srk_chan_stats_chunks(HW))
    part = torch.empty(B, nck, 2, C, device="cuda")
    check(L.srk_chan_stats(pd.data_ptr() + 16, ld, qd.data_ptr() + 16, ld, part.data_ptr(), B, HW, C // 8, _st()))
    got = part.sum(1).cpu()
    pf, qf = p.float()[:, 8:8 + C].reshape(B, HW, C), q.float()[:, 8:8 + C].reshape(B, HW, C)
    assert _rel(got[:, 0], pf.sum(1)) <= 1e-5 and _rel(got[:, 1], (pf * qf).sum(1)) <= 1e-5
    # affine (+ GELU) with per-sample vectors; dgelu_affine; lincomb2
    A, Bv, Cv = torch.randn(B, C, generator=g), torch.randn(B, C, generator=g), torch.randn(B, C, generator=g)
    Ad, Bd, Cd = A.cuda(), Bv.cuda(), Cv.cuda()
    out = torch.zeros(T, ld, dtype=torch.bfloat16, device="cuda")
    check(L.srk_affine_act_bf16(pd.data_ptr() + 16, ld, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), ld, T, C // 8, HW, 1, _st()))
    want = F.gelu(pf * A[:, None] + Bv[:, None]).reshape(T, C)
    assert float((out.cpu().float()[:, :C] - want).abs().max()) <= 1e-2 * float(want.abs().max())
    check(L.srk_dgelu_affine_bf16(qd.data_ptr() + 16, ld, pd.data_ptr() + 16, ld, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), ld, T, C // 8, _st()))
    z = (pf.reshape(T, C) * A[0] + Bv[0]).requires_grad_(True)
    F.gelu(z).backward(qf.reshape(T, C))
    assert _rel(out.cpu().float()[:, :C], z.grad) <= 1e-2
    acc = _bf(torch.randn(T, ld, generator=g))
    accd = acc.cuda()
    check(L.srk_lincomb2_bf16(pd.data_ptr() + 16, ld, qd.data_ptr() + 16, ld, Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), accd.data_ptr(), ld, T,
                              C // 8, HW, 1, _st()))
    want = acc.float()[:, :C] + (pf * A[:, None] + qf * Bv[:, None] + Cv[:, None]).reshape(T, C)
    assert float((accd.cpu().float()[:, :C] - want).abs().max()) <= 1e-2 * float(want.abs().max())
    check(L.srk_lincomb2_bf16(pd.data_ptr() + 16, ld, None, 0, None, None, None, accd.data_ptr(), ld, T, C // 8, 0, 0, _st()))       # plain copy
    assert torch.equal(accd.cpu()[:, :C], p[:, 8:8 + C])
    # mul_bwd
    da, db = torch.zeros(T, ld, dtype=torch.bfloat16, device="cuda"), torch.zeros(T, ld, dtype=torch.bfloat16, device="cuda")
    check(L.srk_mul_bwd_bf16(accd.data_ptr(), ld, pd.data_ptr() + 16, ld, qd.data_ptr() + 16, ld, da.data_ptr(), ld, db.data_ptr(), ld, T, C // 8, _st()))
    dyf = accd.cpu().float()[:, :C]
    assert _rel(da.cpu().float()[:, :C], dyf * qf.reshape(T, C)) <= 5e-3 and _rel(db.cpu().float()[:, :C], dyf * pf.reshape(T, C)) <= 5e-3
    # depth-wise 3x3 weight / bias gradient
    wt = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).requires_grad_(True)
    bs = torch.zeros(C, requires_grad=True)
    xi = pf.reshape(B, H, W, C).permute(0, 3, 1, 2)
    F.conv2d(xi, wt, bs, padding=1, groups=C).backward(qf.reshape(B, H, W, C).permute(0, 3, 1, 2))
    part = torch.empty(B, int(L.srk_dwconv3x3_wgrad_chunks(H)), 10, C, device="cuda")
    check(L.srk_dwconv3x3_wgrad(qd.data_ptr() + 16, ld, pd.data_ptr() + 16, ld, part.data_ptr(), B, H, W, C // 8, _st()))
    got = part.sum((0, 1)).cpu()
    assert _rel(got[:9].t().reshape(C, 1, 3, 3), wt.grad) <= 1e-5 and _rel(got[9], bs.grad) <= 1e-5
    # dual gate backward
    CA = 64
    a1, a2, dc = (_bf(torch.randn(T, CA, generator=g)) for _ in range(3))
    cg, smap = torch.rand(B, CA, generator=g), torch.randn(T, generator=g)
    a1f, a2f = a1.float().requires_grad_(True), a2.float().requires_grad_(True)
    cgf, smf = cg.clone().requires_grad_(True), smap.clone().requires_grad_(True)
    comb = a1f.reshape(B, HW, CA) * cgf[:, None] + a2f.reshape(B, HW, CA) * torch.sigmoid(smf).reshape(B, HW, 1)
    comb.backward(dc.float().reshape(B, HW, CA))
    tg = torch.sigmoid(smap).cuda()
    d1, d2 = torch.empty(T, CA, dtype=torch.bfloat16, device="cuda"), torch.empty(T, CA, dtype=torch.bfloat16, device="cuda")
    dcgp, dsm = torch.empty(B, (HW + 63) // 64, CA, device="cuda"), torch.empty(T, device="cuda")
    dcd, a1d, a2d, cgd = dc.cuda(), a1.cuda(), a2.cuda(), cg.cuda()
    check(L.srk_dual_gate_bwd(dcd.data_ptr(), a1d.data_ptr(), a2d.data_ptr(), cgd.data_ptr(), tg.data_ptr(), d1.data_ptr(), d2.data_ptr(),
                              dcgp.data_ptr(), dsm.data_ptr(), B, HW, CA, _st()))
    assert _rel(d1.cpu().float(), a1f.grad) <= 5e-3 and _rel(d2.cpu().float(), a2f.grad) <= 5e-3
    assert _rel(dcgp.sum(1).cpu(), cgf.grad) <= 1e-5 and _rel(dsm.cpu(), smf.grad) <= 1e-5
    # LayerNorm backward on a bf16 row slice
    rows, Cn, CPn = 300, 90, 128          # C not a multiple of 8: the last 16-byte piece straddles C
    h = _bf(torch.randn(rows, 256, generator=g))
    dyl = _bf(torch.randn(rows, CPn, generator=g))
    gm = torch.rand(Cn, generator=g) + 0.5
    hx = h.float()[:, 128:128 + Cn].requires_grad_(True)
    gmr, btr = gm.clone().requires_grad_(True), torch.zeros(Cn, requires_grad=True)
    F.layer_norm(hx, (Cn,), gmr, btr, 1e-5).backward(dyl.float()[:, :Cn])
    dxl = torch.full((rows, 256), 7.0, dtype=torch.bfloat16, device="cuda")
    nb = int(L.srk_rowln_bwd_blocks(rows))
    part = torch.empty(nb, 2, Cn, device="cuda")
    dyd, hd_, gmd = dyl.cuda(), h.cuda(), gm.cuda()
    check(L.srk_rowln_bwd_bf16(dyd.data_ptr(), CPn, hd_.data_ptr() + 128 * 2, 256, gmd.data_ptr(), dxl.data_ptr() + 128 * 2, 256, part.data_ptr(), rows,
                               Cn, CPn, _st()))
    got = dxl.cpu().float()
    assert _rel(got[:, 128:128 + Cn], hx.grad) <= 1e-2 and float(got[:, 128 + Cn:].abs().max()) == 0.0 and float(got[:, :128].min()) == 7.0
    ps = part.sum(0).cpu()
    assert _rel(ps[0], gmr.grad) <= 1e-4 and _rel(ps[1], btr.grad) <= 1e-4


def test_dat_spatial_gate_train_passes_vs_autograd():
    """The three passes of the spatial interaction in training (statistics, backward statistics, backward apply) with the closed-form
    BatchNorm coefficients of dat_train.py against torch autograd of conv1x1 -> BatchNorm(train) -> GELU -> conv1x1."""
    from tpu_superresolution_amd import dat_train as DT
    check, L = _lib()
    g = torch.Generator().manual_seed(2)
    T, CA, S = 700, 128, 7
    x = _bf(torch.randn(T, CA, generator=g))
    W0, b0 = torch.randn(S, CA, generator=g) * 0.1, torch.randn(S, generator=g) * 0.1
    gam, bet, w3 = torch.rand(S, generator=g) + 0.5, torch.randn(S, generator=g) * 0.1, torch.randn(S, generator=g)
    dsm = torch.randn(T, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x.float(), W0, b0, gam, bet, w3)]
    xf, W0r, b0r, gr, br, w3r = leaves
    y1 = xf @ W0r.t() + b0r
    z = (y1 - y1.mean(0)) / torch.sqrt(y1.var(0, unbiased=False) + 1e-5) * gr + br
    smap = F.gelu(z) @ w3r
    smap.backward(dsm)
    xd, W0d, b0d, w3d, dsd = x.cuda(), W0.cuda(), b0.cuda(), w3.cuda(), dsm.cuda()
    nblk = (T + 255) // 256
    part = torch.empty(nblk, 2, 16, device="cuda")
    check(L.srk_spatial_gate_train(0, xd.data_ptr(), CA, W0d.data_ptr(), b0d.data_ptr(), None, None, None, None, None, None, None, None, 0, 0,
                                   part.data_ptr(), T, CA, S, _st()))
    ps = part.sum(0)
    s_, t_, mu, rstd, var = DT._bn_coeffs(ps[0, :S], ps[1, :S], T, gam.cuda(), bet.cuda())
    assert _rel(mu.cpu(), y1.detach().mean(0)) <= 1e-4 and _rel(var.cpu(), y1.detach().var(0, unbiased=False)) <= 1e-4
    # the one-launch coefficient kernel against the closed form above, running buffers against nn.BatchNorm's own update
    gd, bd = gam.cuda(), bet.cuda()
    coef = torch.full((4, 16), 7.0, device="cuda")
    rm, rv = torch.randn(S, generator=g).cuda(), (torch.rand(S, generator=g) + 0.5).cuda()
    bn_ref = torch.nn.BatchNorm1d(S, momentum=0.1)
    bn_ref.running_mean.copy_(rm.cpu()); bn_ref.running_var.copy_(rv.cpu())
    bn_ref.train()(y1.detach())
    check(L.srk_bn_train_coeffs(part.data_ptr(), nblk, 32, 16, S, float(T), gd.data_ptr(), bd.data_ptr(), 1e-5, coef.data_ptr(), rm.data_ptr(),
                                rv.data_ptr(), 0.1, None, _st()))
    for row, want in enumerate((s_, t_, mu, rstd)):
        assert _rel(coef[row, :S].cpu(), want.cpu()) <= 1e-5, row
    assert _rel(rm.cpu(), bn_ref.running_mean) <= 1e-4 and _rel(rv.cpu(), bn_ref.running_var) <= 1e-4
    fwd_coef = coef
    part = torch.empty(nblk, 4, 16, device="cuda")
    check(L.srk_spatial_gate_train(1, xd.data_ptr(), CA, W0d.data_ptr(), b0d.data_ptr(), s_.data_ptr(), t_.data_ptr(), w3d.data_ptr(), dsd.data_ptr(),
                                   None, None, None, None, 0, 0, part.data_ptr(), T, CA, S, _st()))
    ps = part.sum(0)
    cA, cB, cC, dgam, dbet = (v.contiguous() for v in DT._bn_backward_coeffs(ps[0, :S], ps[1, :S], T, s_, mu, rstd))
    assert _rel(dgam.cpu(), gr.grad) <= 1e-3 and _rel(dbet.cpu(), br.grad) <= 1e-3 and _rel(ps[2, :S].cpu(), w3r.grad) <= 1e-3
    bc = torch.full((5, 16), 7.0, device="cuda")
    check(L.srk_bn_train_bwd_coeffs(part.data_ptr(), nblk, 64, 16, S, float(T), fwd_coef.data_ptr(), bc.data_ptr(), _st()))
    for row, want in enumerate((cA, cB, cC, dgam, dbet)):
        assert _rel(bc[row, :S].cpu(), want.cpu()) <= 1e-4, row
    assert abs(float(ps[3, 0]) - float(dsm.sum())) <= 1e-3
    dx = torch.zeros(T, CA, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(nblk, 16, CA + 1, device="cuda")
    check(L.srk_spatial_gate_train(2, xd.data_ptr(), CA, W0d.data_ptr(), b0d.data_ptr(), s_.data_ptr(), t_.data_ptr(), w3d.data_ptr(), dsd.data_ptr(),
                                   cA.data_ptr(), cB.data_ptr(), cC.data_ptr(), dx.data_ptr(), CA, 0, part.data_ptr(), T, CA, S, _st()))
    ps = part.view(nblk, -1).sum(0).cpu()
    assert _rel(dx.cpu().float(), xf.grad) <= 1e-2
    assert _rel(ps[:16 * CA].view(16, CA)[:S], W0r.grad) <= 2e-3
    assert float(ps[16 * CA:16 * CA + S].abs().max()) <= 1e-3 * float(dsm.abs().sum())        # the bias in front of a BatchNorm gets no gradient


def test_dat_channel_attention_train_path_vs_autograd():
    """chan_gram + the d x d softmax of dat_train.py + chan_apply_mat, forward and backward, against autograd of the reference's
    formulation (:497-505) on the same bf16 q / k / v."""
    from tpu_superresolution_amd import dat_train as DT
    check, L = _lib()
    g = torch.Generator().manual_seed(3)
    B, N, nH, dh = 2, 600, 4, 12
    CA = nH * 32
    qkv = torch.zeros(B * N, 3 * CA)
    real = torch.randn(B * N, 3, nH, dh, generator=g)
    qkv.view(B * N, 3, nH, 32)[..., :dh] = real
    qkv = _bf(qkv)
    temp = torch.rand(nH, 1, 1, generator=g) + 0.5
    dout = _bf(torch.randn(B * N, CA, generator=g))
    r = qkv.float().view(B, N, 3, nH, 32)[..., :dh].permute(2, 0, 3, 4, 1).clone().requires_grad_(True)       # [3][B][h][d][N]
    tr = temp.clone().requires_grad_(True)
    attn = ((F.normalize(r[0], dim=-1) @ F.normalize(r[1], dim=-1).transpose(-2, -1)) * tr).softmax(-1)
    out = (attn @ r[2]).permute(0, 3, 1, 2)                                                                    # [B][N][h][d]
    out.backward(dout.float().view(B, N, nH, 32)[..., :dh])
    qd = qkv.cuda()
    part = torch.empty(int(L.srk_chan_gram_floats(B, N, nH)), device="cuda")
    check(L.srk_chan_gram(qd.data_ptr(), 3 * CA, qd.data_ptr() + CA * 2, 3 * CA, part.data_ptr(), B, N, nH, _st()))
    gg = part.view(B, nH, -1, 1088).sum(2)
    G, sq, sk = gg[..., :1024].reshape(B, nH, 32, 32), gg[..., 1024:1056], gg[..., 1056:]
    A_t = DT._channel_attention_matrix(G, sq, sk, temp.cuda(), dh).contiguous()
    # the one-launch form of the same function (csrc/dat_small.hip): chunk partials -> gram, A
    tempd = temp.cuda().contiguous()
    nchunk = part.numel() // (B * nH * 1088)
    gram, A = torch.empty(B, nH, 1088, device="cuda"), torch.full((B, nH, 32, 32), 7.0, device="cuda")
    check(L.srk_chan_attn_matrix_fwd(part.data_ptr(), nchunk, tempd.data_ptr(), gram.data_ptr(), A.data_ptr(), B, nH, dh, _st()))
    assert _rel(gram.cpu(), gg.cpu()) <= 1e-6
    assert float((A - A_t).abs().max()) <= 1e-6
    assert float((A[..., :dh, :dh].cpu() - attn.detach()).abs().max()) <= 1e-5
    att = torch.zeros(B * N, CA, dtype=torch.bfloat16, device="cuda")
    check(L.srk_chan_apply_mat(A.data_ptr(), qd.data_ptr() + 2 * CA * 2, 3 * CA, None, None, 0, att.data_ptr(), CA, B, N, nH, 0, _st()))
    got = att.cpu().float().view(B, N, nH, 32)
    assert _rel(got[..., :dh], out.detach()) <= 5e-3 and float(got[..., dh:].abs().max()) == 0.0
    dod = dout.cuda()
    check(L.srk_chan_gram(dod.data_ptr(), CA, qd.data_ptr() + 2 * CA * 2, 3 * CA, part.data_ptr(), B, N, nH, _st()))
    dA = part.view(B, nH, -1, 1088).sum(2)[..., :1024].reshape(B, nH, 32, 32)
    dqkv = torch.zeros(B * N, 3 * CA, dtype=torch.bfloat16, device="cuda")
    At = A.transpose(-1, -2).contiguous()
    check(L.srk_chan_apply_mat(At.data_ptr(), dod.data_ptr(), CA, None, None, 0, dqkv.data_ptr() + 2 * CA * 2, 3 * CA, B, N, nH, 0, _st()))
    Gm, sqm, skm, tm = (t.detach().requires_grad_(True) for t in (G, sq, sk, temp.cuda()))
    dG, dsq, dsk, dtemp = torch.autograd.grad(DT._channel_attention_matrix(Gm, sqm, skm, tm, dh), [Gm, sqm, skm, tm], dA)
    dG_t, dsq2_t, dsk2_t = dG.contiguous(), (2 * dsq).contiguous(), (2 * dsk).contiguous()
    dG, dGt = torch.full((B, nH, 32, 32), 7.0, device="cuda"), torch.full((B, nH, 32, 32), 7.0, device="cuda")
    dsq2, dsk2, dtemp_bh = torch.full((B, nH, 32), 7.0, device="cuda"), torch.full((B, nH, 32), 7.0, device="cuda"), torch.empty(B, nH, device="cuda")
    check(L.srk_chan_attn_matrix_bwd(part.data_ptr(), nchunk, gram.data_ptr(), A.data_ptr(), tempd.data_ptr(), dG.data_ptr(), dGt.data_ptr(),
                                     dsq2.data_ptr(), dsk2.data_ptr(), dtemp_bh.data_ptr(), B, nH, dh, _st()))
    assert _rel(dG.cpu(), dG_t.cpu()) <= 1e-4 and torch.equal(dGt, dG.transpose(-1, -2))
    assert _rel(dsq2.cpu(), dsq2_t.cpu()) <= 1e-4 and _rel(dsk2.cpu(), dsk2_t.cpu()) <= 1e-4
    assert _rel(dtemp_bh.sum(0).cpu(), dtemp.flatten().cpu()) <= 1e-4
    dtemp = dtemp_bh.sum(0).view(nH, 1, 1)
    check(L.srk_chan_apply_mat(dG.data_ptr(), qd.data_ptr() + CA * 2, 3 * CA, dsq2.data_ptr(), qd.data_ptr(), 3 * CA, dqkv.data_ptr(), 3 * CA, B, N, nH, 0,
                               _st()))
    check(L.srk_chan_apply_mat(dGt.data_ptr(), qd.data_ptr(), 3 * CA, dsk2.data_ptr(), qd.data_ptr() + CA * 2, 3 * CA, dqkv.data_ptr() + CA * 2, 3 * CA, B,
                               N, nH, 0, _st()))
    got = dqkv.cpu().float().view(B, N, 3, nH, 32)[..., :dh].permute(2, 0, 3, 4, 1)
    for w_, nm in enumerate("qkv"):
        assert _rel(got[w_], r.grad[w_]) <= 2e-2, nm
    assert _rel(dtemp.cpu(), tr.grad) <= 1e-3


@pytest.mark.parametrize("B,C,S,nH", [(16, 180, 22, 6), (3, 60, 7, 4), (5, 24, 3, 2)])
def test_dat_channel_interaction_kernels_vs_autograd(B, C, S, nH):
    """srk_channel_interaction_fwd / _bwd (csrc/dat_small.hip) against dat_train._channel_interaction under autograd (the formulation of
    dat_arch.py:315-321 on a 1 x 1 map with BatchNorm batch statistics): gate, running buffers, all six parameter gradients and the
    pooled gradient, through the head-padded channel layout."""
    from tpu_superresolution_amd import dat_train as DT
    from tpu_superresolution_amd import hat_arch as ha
    check, L = _lib()
    g = torch.Generator().manual_seed(B * 1000 + C)
    dh = C // nH
    CA, HW = nH * 32, 64
    ci = torch.nn.Sequential(torch.nn.AdaptiveAvgPool2d(1), torch.nn.Conv2d(C, S, 1), torch.nn.BatchNorm2d(S), torch.nn.GELU(),
                             torch.nn.Conv2d(S, C, 1)).cuda()
    with torch.no_grad():
        for p_ in ci.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.5)
        ci[2].running_mean.copy_(torch.randn(S, generator=g))
        ci[2].running_var.copy_(torch.rand(S, generator=g) + 0.5)
    rm0, rv0 = ci[2].running_mean.clone(), ci[2].running_var.clone()
    hm = ha._head_map(nH, dh, torch.device("cuda"))
    pooled = torch.zeros(B, 2, CA, device="cuda")
    pooled[:, 0, hm] = (torch.randn(B, C, generator=g) * HW).cuda()
    pooled[:, 1] = 99.0                                                   # the second row of the partial sums is not read
    dcg = torch.zeros(B, CA, device="cuda")
    dcg[:, hm] = torch.randn(B, C, generator=g).cuda()
    # torch path
    pm_t = (pooled[:, 0] / HW)[:, hm].contiguous().requires_grad_(True)
    st_: dict = {}
    cg_t = DT._channel_interaction(pm_t, ci, st_)
    params = [ci[1].weight, ci[1].bias, ci[2].weight, ci[2].bias, ci[4].weight, ci[4].bias]
    grads = torch.autograd.grad(cg_t, [pm_t] + params, dcg[:, hm])
    DT._bn_update(ci[2], st_["mean"], st_["var"], st_["n"])
    rm_t, rv_t = ci[2].running_mean.clone(), ci[2].running_var.clone()
    ci[2].running_mean.copy_(rm0); ci[2].running_var.copy_(rv0)
    # kernels
    assert DT._ci_fused_ok(B, C, S, ci)
    hm32 = hm.to(torch.int32)
    pm, cgate = torch.empty(B, C, device="cuda"), torch.full((B, CA), 7.0, device="cuda")
    view = pooled[:, 0]
    check(L.srk_channel_interaction_fwd(view.data_ptr(), view.stride(0), 1.0 / HW, hm32.data_ptr(), ci[1].weight.data_ptr(), ci[1].bias.data_ptr(),
                                        ci[2].weight.data_ptr(), ci[2].bias.data_ptr(), float(ci[2].eps), ci[4].weight.data_ptr(), ci[4].bias.data_ptr(),
                                        ci[2].running_mean.data_ptr(), ci[2].running_var.data_ptr(), float(ci[2].momentum), pm.data_ptr(),
                                        cgate.data_ptr(), B, C, S, CA, _st()))
    assert _rel(pm.cpu(), pm_t.detach().cpu()) <= 1e-6
    assert float((cgate[:, hm] - cg_t.detach()).abs().max()) <= 2e-5
    pad = torch.ones(CA, dtype=torch.bool, device="cuda"); pad[hm] = False
    assert float(cgate[:, pad].abs().max()) == 0.0
    assert _rel(ci[2].running_mean.cpu(), rm_t.cpu()) <= 1e-5 and _rel(ci[2].running_var.cpu(), rv_t.cpu()) <= 1e-5
    gk = [torch.full(p_.shape, 7.0, device="cuda") for p_ in params]
    dpool = torch.full((B, CA), 7.0, device="cuda")
    check(L.srk_channel_interaction_bwd(pm.data_ptr(), dcg.data_ptr(), CA, 1.0 / HW, hm32.data_ptr(), ci[1].weight.data_ptr(), ci[1].bias.data_ptr(),
                                        ci[2].weight.data_ptr(), ci[2].bias.data_ptr(), float(ci[2].eps), ci[4].weight.data_ptr(), ci[4].bias.data_ptr(),
                                        gk[0].data_ptr(), gk[1].data_ptr(), gk[2].data_ptr(), gk[3].data_ptr(), gk[4].data_ptr(), gk[5].data_ptr(),
                                        dpool.data_ptr(), B, C, S, CA, _st()))
    for nm, got, want in zip(("W1", "b1", "gamma", "beta", "W2", "b2"), gk, grads[1:]):
        if nm == "b1":                   # a bias in front of a BatchNorm: exactly zero in exact arithmetic, rounding noise in both
            assert float(got.abs().max()) <= 1e-4 * max(1.0, float(grads[1].abs().max())), nm
        else:
            assert _rel(got.cpu(), want.cpu()) <= 2e-4, nm
    assert _rel(dpool[:, hm].cpu(), (grads[0] / HW).cpu()) <= 2e-4 and float(dpool[:, pad].abs().max()) == 0.0


@pytest.mark.parametrize("wh,ww,shift,H,W", [(8, 32, True, 32, 64), (32, 8, False, 24, 40), (8, 16, True, 24, 40), (16, 8, True, 32, 32),
                                             (16, 16, True, 32, 48)])          # 16 x 16: HAT's W-MSA backward runs through this kernel
def test_dat_rect_window_attention_backward_vs_autograd(wh, ww, shift, H, W):
    """csrc/attn_rect_bwd.hip against autograd of the oracle's window attention on the padded frame (zero-padded q / k / v, cyclic shift,
    arithmetic mask, dense bias): d q / d k / d v per token and the dense bias gradient."""
    check, L = _lib()
    g = torch.Generator().manual_seed(wh * 100 + ww + H)
    B, nH, dh = 2, 2, 12
    CA = 2 * nH * 32                    # the launch covers one branch's half of the heads (as Adaptive_Spatial_Attention does)
    N = wh * ww
    big = max(wh, ww)
    Hp, Wp = (H + big - 1) // big * big, (W + big - 1) // big * big
    sy, sx = (wh // 2, ww // 2) if shift else (0, 0)
    scale = dh ** -0.5
    T = B * H * W
    qkv = torch.zeros(T, 3, 2 * nH, 32)
    qkv[..., :nH, :dh] = torch.randn(T, 3, nH, dh, generator=g)
    qkv = _bf(qkv.reshape(T, 3 * CA))
    bias = torch.randn(nH, N, N, generator=g) * 0.5
    dout = torch.zeros(T, 2 * nH, 32)
    dout[:, :nH, :dh] = torch.randn(T, nH, dh, generator=g)
    dout = _bf(dout.reshape(T, CA))
    # reference: the oracle's formulation with autograd
    x = qkv.float().view(B, H, W, 3, 2 * nH, 32)[..., :nH, :dh].clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    xp = F.pad(x, (0, 0, 0, 0, 0, 0, 0, Wp - W, 0, Hp - H)).reshape(B, Hp * Wp, 3, nH, dh)
    idx = torch.from_numpy(DO.rect_window_token_index(Hp, Wp, wh, ww, sy, sx))
    nW = idx.shape[0]
    win = xp[:, idx.reshape(-1)].reshape(B * nW, N, 3, nH, dh).permute(2, 0, 3, 1, 4)
    attn = (win[0] * scale) @ win[1].transpose(-2, -1) + br[None]
    if shift:
        attn = (attn.reshape(B, nW, nH, N, N) + torch.from_numpy(DO.rect_shift_mask(Hp, Wp, wh, ww, sy, sx))[None, :, None]).reshape(-1, nH, N, N)
    o = (attn.softmax(-1) @ win[2]).transpose(1, 2).reshape(B, nW * N, nH, dh)
    merged = torch.zeros(B, Hp * Wp, nH, dh).index_copy(1, idx.reshape(-1), o).reshape(B, Hp, Wp, nH, dh)[:, :H, :W]
    merged.backward(dout.float().view(B, H, W, 2 * nH, 32)[..., :nH, :dh])
    qd, dod, bd = qkv.cuda(), dout.cuda(), bias.cuda()
    dqkv = torch.zeros(T, 3 * CA, dtype=torch.bfloat16, device="cuda")
    dbias, dbias2 = torch.zeros(nH, N, N, device="cuda"), torch.zeros(nH, N, N, device="cuda")
    scr = torch.empty(int(L.srk_win_attention_bwd_padded_scratch(B, Hp, Wp, wh, ww, nH)), dtype=torch.uint8, device="cuda")
    check(L.srk_win_attention_bwd_padded(qd.data_ptr(), 3 * CA, CA, bd.data_ptr(), dod.data_ptr(), CA, dqkv.data_ptr(), dbias2.data_ptr(), None, B, H, W,
                                         Hp, Wp, wh, ww, sy, sx, nH, scale, _st()))          # without scratch: float atomics
    dqkv.zero_()
    check(L.srk_win_attention_bwd_padded(qd.data_ptr(), 3 * CA, CA, bd.data_ptr(), dod.data_ptr(), CA, dqkv.data_ptr(), dbias.data_ptr(), scr.data_ptr(),
                                         B, H, W, Hp, Wp, wh, ww, sy, sx, nH, scale, _st()))  # per-window tiles + reduction
    assert _rel(dbias2.cpu(), dbias.cpu()) <= 1e-5
    got = dqkv.cpu().float().view(B, H, W, 3, 2 * nH, 32)
    for w_, nm in enumerate("qkv"):
        assert _rel(got[..., w_, :nH, :dh], x.grad[..., w_, :, :]) <= 2e-2, nm
    assert float(got[..., nH:, :].abs().max()) == 0.0                 # the other branch's heads are not this launch's
    assert _rel(dbias.cpu(), br.grad) <= 1e-2


def _train_model(cfg, sd, drop_path_rate=0.0):
    import tpu_superresolution_amd as T
    m = T.DAT(**cfg.kwargs(), drop_path_rate=drop_path_rate)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def _check_grads(m, want: dict, tol=0.1, floor=2e-3):
    """per-tensor relative error <= tol, measured against max(|reference gradient|, floor * the largest gradient norm of the model): the
    gradients that are mathematically zero (a bias in front of a BatchNorm, a softmax-invariant shift) are noise in both."""
    biggest = max(float(v.norm()) for v in want.values())
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        w = want[n]
        e = float((p.grad.cpu().float() - w).norm()) / max(float(w.norm()), floor * biggest)
        if e > worst[1]:
            worst = (n, e)
    print(f"worst gradient error {worst[1]:.3e} at {worst[0]}")
    assert worst[1] <= tol, worst


@pytest.mark.parametrize("tag", ["32x32", "24x40"])
def test_dat_train_step_vs_reference_golden(tag):
    """G14c: one training step of the reference's DAT in .train() (BatchNorm batch statistics, drop_path 0): loss, output, every
    parameter's gradient and the BatchNorm buffers after the step."""
    from test_oracle_golden import DAT_TINY
    g = load_golden("g14c_dat_train")
    cfg = DO.DATConfig(**DAT_TINY)
    sd = DO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    m = _train_model(cfg, sd)
    x, t = torch.from_numpy(g[f"{tag}.x"]).cuda(), torch.from_numpy(g[f"{tag}.t"]).cuda()
    y = m(x)
    loss = F.l1_loss(y, t)
    loss.backward()
    yr = torch.from_numpy(g[f"{tag}.y"])
    assert float((y.detach().cpu() - yr).abs().max()) <= 2e-2 * float(yr.abs().max())
    assert abs(float(loss) - float(g[f"{tag}.loss"])) <= 5e-3 * float(g[f"{tag}.loss"])
    _check_grads(m, {n: torch.from_numpy(g[f"{tag}.grad.{n}"]) for n, _ in m.named_parameters()})
    for n, b in m.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            r = torch.from_numpy(g[f"{tag}.buf.{n}"])
            assert float((b.cpu() - r).abs().max()) <= 2e-2 * max(float(r.abs().max()), 1e-2), n
        elif n.endswith("num_batches_tracked"):
            assert int(b) == int(g[f"{tag}.buf.{n}"])
    # the eval forward afterwards folds the UPDATED running statistics
    m.eval()
    with torch.no_grad():
        ye = m(x)
    sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = DO.dat_forward(sd2, cfg, x.cpu())
    assert float((ye.cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())


def test_dat_train_step_with_drop_path_vs_oracle():
    """DropPath factors drawn by DAT.forward (seeded) and handed to the oracle: loss and gradients of the same step."""
    from test_oracle_golden import DAT_TINY_816
    cfg = DO.DATConfig(**DAT_TINY_816)
    sd = DO.random_state_dict(cfg, seed=21, scale=2.0)
    m = _train_model(cfg, sd, drop_path_rate=0.3)
    gen = torch.Generator().manual_seed(4)
    x, t = torch.rand(3, 3, 24, 40, generator=gen), torch.rand(3, 3, 48, 80, generator=gen)
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    y = m(x.cuda())
    F.l1_loss(y, t.cuda()).backward()
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    probs = [blk.drop_path_prob for layer in m.layers for blk in layer.blocks]
    keep = 1.0 - torch.tensor(probs, dtype=torch.float32, device="cuda").view(-1, 1, 1)
    drop = ((torch.rand(len(probs), 2, 3, device="cuda") < keep).float() / keep).cpu()
    assert float(drop.min()) == 0.0                                   # some branch is dropped for some sample
    loss, yo, grads, _ = DO.loss_and_grads(sd, cfg, x, t, drop)
    assert float((y.detach().cpu() - yo).abs().max()) <= 2e-2 * float(yo.abs().max())
    _check_grads(m, grads)


def test_dat_train_step_pixelshuffledirect_vs_oracle():
    """The light-weight head (UpsampleOneStep, dat_arch.py:846-848) in training: fp32 small-conv gradients through the un-shuffled image."""
    from test_oracle_golden import DAT_TINY
    cfg = DO.DATConfig(**dict(DAT_TINY, upsampler="pixelshuffledirect", depth=(2,), num_heads=(4,)))
    sd = DO.random_state_dict(cfg, seed=23, scale=2.0)
    m = _train_model(cfg, sd)
    gen = torch.Generator().manual_seed(6)
    x, t = torch.rand(2, 3, 32, 32, generator=gen), torch.rand(2, 3, 64, 64, generator=gen)
    y = m(x.cuda())
    F.l1_loss(y, t.cuda()).backward()
    loss, yo, grads, _ = DO.loss_and_grads(sd, cfg, x, t)
    assert float((y.detach().cpu() - yo).abs().max()) <= 2e-2 * float(yo.abs().max())
    _check_grads(m, grads)


def test_graphed_train_step_matches_eager_steps():
    """training.GraphedTrainStep (the whole step -- pack, forward, L1, backward, clip, AdamW -- as one hipGraph replay) against the same
    steps launched eagerly: losses and weights after four steps on changing batches (drop_path 0; fp32 atomics in a few reductions are
    the only difference)."""
    from test_oracle_golden import DAT_TINY
    from tpu_superresolution_amd.training import GraphedTrainStep, l1_loss_checked
    cfg = DO.DATConfig(**DAT_TINY)
    sd = DO.random_state_dict(cfg, seed=31, scale=1.0)
    gen = torch.Generator().manual_seed(9)
    batches = [(torch.rand(2, 3, 32, 32, generator=gen).cuda(), torch.rand(2, 3, 64, 64, generator=gen).cuda()) for _ in range(4)]
    ma, mb = _train_model(cfg, sd), _train_model(cfg, sd)
    oa = torch.optim.AdamW(ma.parameters(), lr=1e-3, weight_decay=0.0)
    ob = torch.optim.AdamW(mb.parameters(), lr=1e-3, weight_decay=0.0, capturable=True)
    gs = GraphedTrainStep(mb, ob, max_grad_norm=1.0, warmup=1)
    # the graphed stepper warms up with one eager step on its first batch: give the eager model the same extra step
    oa.zero_grad(set_to_none=True)
    l0, _ = l1_loss_checked(ma(batches[0][0]), batches[0][1])
    l0.backward()
    torch.nn.utils.clip_grad_norm_(ma.parameters(), 1.0)
    oa.step()
    la, lb = [], []
    for x, t in batches:
        oa.zero_grad(set_to_none=True)
        loss, _ = l1_loss_checked(ma(x), t)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ma.parameters(), 1.0)
        oa.step()
        la.append(float(loss))
        lg, bad = gs(x, t)
        lb.append(float(lg))
        assert int(bad) == 0
    print("eager", la, "graphed", lb)
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(la, lb))
    assert lb[-1] < lb[0]
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        # Adam normalises the update: an element whose tiny gradient flips sign between the two runs moves by up to 2 lr per step, and the
        # parameters whose true gradient is zero (a bias in front of a BatchNorm, a softmax-invariant shift) random-walk on rounding noise
        if n.endswith(("dwconv.0.bias", "channel_interaction.1.bias", "spatial_interaction.0.bias", "pos3.2.bias")):
            continue
        assert float((pa - pb).abs().max()) <= 4e-3 and float((pa - pb).abs().mean()) <= 2e-4, n
    for (n, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        if n.endswith("num_batches_tracked"):
            assert int(ba) == int(bb) == 5


def test_dat_width_180_train_step_vs_oracle():
    """The official DAT width (embed 180, 6 heads, split 8x32, expansion 4; two blocks: one spatial, one channel) with 16 384 tokens: here the
    persistent streaming GEMMs run -- incl. the qkv dgrad with the norm1 backward in its epilogue (SRK_EP_LNBWD in token order) -- where
    the tiny-model tests take the tile kernels.  Loss, output and every gradient against the training oracle."""
    cfg = DO.DATConfig(**{**DO.DATConfig.sr_x4().__dict__, "depth": (2,), "num_heads": (6,), "upscale": 2})
    sd = DO.random_state_dict(cfg, seed=41, scale=1.0)
    m = _train_model(cfg, sd)
    gen = torch.Generator().manual_seed(12)
    x, t = torch.rand(4, 3, 64, 64, generator=gen), torch.rand(4, 3, 128, 128, generator=gen)
    y = m(x.cuda())
    loss = F.l1_loss(y, t.cuda())
    loss.backward()
    lo, yo, grads, rec = DO.loss_and_grads(sd, cfg, x, t)
    assert float((y.detach().cpu() - yo).abs().max()) <= 2e-2 * float(yo.abs().max())
    assert abs(float(loss.detach()) - lo) <= 5e-3 * lo
    _check_grads(m, grads)
    for n, b in m.named_buffers():
        if n.endswith("running_var"):
            assert float((b.cpu() - rec[n]).abs().max()) <= 2e-2 * max(float(rec[n].abs().max()), 1e-2), n


@pytest.mark.parametrize("outer,R,n", [(1, 256, 16 * 257), (16, 64, 256), (1, 7, 5), (3, 33, 70)])
def test_dat_sum_rows_kernel_vs_torch(outer, R, n):
    """srk_sum_rows_f32 (the finishing sum of the per-chunk partial rows of the token passes) against torch.sum, incl. ragged sizes."""
    check, L = _lib()
    g = torch.Generator().manual_seed(outer * 100 + R)
    x = torch.randn(outer, R, n, generator=g).cuda()
    out = torch.full((outer, n), 7.0, device="cuda")
    check(L.srk_sum_rows_f32(x.data_ptr(), outer, R, n, out.data_ptr(), _st()))
    want = x.double().sum(1)
    assert float((out.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("rows,Cn,CPo,ld", [(300, 360, 384, 768), (1000, 180, 192, 384), (77, 90, 96, 192), (5, 30, 32, 64)])
def test_dat_row_layernorm_all_lane_groupings(rows, Cn, CPo, ld):
    """srk_rowln_bf16 with 64 / 32 / 16 lanes per row (C <= 512 / 256 / 128), ragged row counts, column slices, zero padding up to CP_out."""
    check, L = _lib()
    g = torch.Generator().manual_seed(rows + Cn)
    h = torch.randn(rows, ld, generator=g).to(torch.bfloat16)
    gm, bt = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1
    off = ld - CPo                                            # a column slice that ends at the row's end (16-byte aligned)
    want = F.layer_norm(h.float()[:, off:off + Cn], (Cn,), gm, bt, 1e-5)
    hd, gd, bd = h.cuda(), gm.cuda(), bt.cuda()
    o = torch.full((rows, CPo), 7.0, dtype=torch.bfloat16, device="cuda")
    check(L.srk_rowln_bf16(hd.data_ptr() + off * 2, ld, gd.data_ptr(), bd.data_ptr(), o.data_ptr(), CPo, rows, Cn, CPo, _st()))
    got = o.cpu().float()
    assert float((got[:, :Cn] - want).abs().max()) <= 3e-2 and float(got[:, Cn:].abs().max()) == 0.0
