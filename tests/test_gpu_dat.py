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
    B, H, W, C = 2, 12, 24, 40                       # 5 groups of 8 channels; W a multiple of the 8-pixel strip
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
    with pytest.raises(NotImplementedError, match="eval mode only"):
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
