"""Whole-model parity of the HIP SwinIR path against the reference's golden vectors and the CPU oracle.

Arithmetic contract of the HIP path: bf16 MFMA operands, fp32 accumulation, fp32 residual stream,
fp32 LayerNorm statistics and softmax.  Tolerances (stated against the fp32 reference/oracle):
  forward   max|err| <= 1.2e-2 * max|ref|  and mutual PSNR >= 50 dB on image-range outputs
            (the reference's own bf16-autocast forward differs from its fp32 forward by 0.5 % of the
            output range; measured here: 0.3 - 0.6 %)
  gradients per-tensor relative L2 error <= 0.1, median <= 0.04, loss relative error <= 2e-3
  PSNR      |PSNR(hip, target) - PSNR(oracle, target)| <= 0.01 dB   (BASELINE metric gate)
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import swinir_oracle as O
from test_oracle_golden import VARIANTS, tiny_weights

pytestmark = pytest.mark.gpu


def build(cfg, sd, train=False, drop_path_rate=0.0):
    import tpu_superresolution_amd as T
    m = T.SwinIR(drop_path_rate=drop_path_rate, **cfg.kwargs())
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m = m.cuda()
    return m.train() if train else m.eval()


def mutual_psnr(a, b):
    return 10 * np.log10(1.0 / max(float(((a.float() - b.float()) ** 2).mean()), 1e-20))


@pytest.mark.parametrize("tag", ["ps4", "psd2", "ps3", "nc4", "dn1", "ps2_3conv_gray"])
def test_tiny_forward_vs_reference_golden(tag):
    g, cfg, sd = tiny_weights(tag)
    m = build(cfg, sd)
    for hw in ((16, 16), (13, 19), (24, 32)):     # exact multiple / reflect-pad + crop + dynamic mask / non-square
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = m(x.cuda()).cpu()
        ref = torch.from_numpy(g[f"y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape and y.dtype == torch.float32
        err = float((y - ref).abs().max())
        # '3conv' squeezes through C/4 = 8 channels twice per RSTB: fewer terms per sum, so bf16 rounding averages out less
        # (1.1 - 1.3 % measured; the bf16 emulation agrees to 6e-3, tests/test_gpu_emulation.py)
        tol = 2e-2 if cfg.resi_connection == "3conv" else 1.2e-2
        assert err <= tol * float(ref.abs().max()), f"{tag} {hw}: max err {err:.3e} vs ref max {float(ref.abs().max()):.3e}"
    # the module still round-trips the reference state_dict exactly
    out_sd = m.state_dict()
    assert list(out_sd.keys()) == list(sd.keys())
    for k, v in sd.items():
        assert torch.equal(out_sd[k].cpu(), v), k


@pytest.mark.parametrize("tag,cfg,hw", [("cfg2", O.SwinIRConfig.light_x2(), 48), ("cfg3", O.SwinIRConfig.classical_x4(), 64)])
def test_full_size_forward_probes_and_psnr(tag, cfg, hw):
    g = load_golden(f"g10_{tag}_probe")
    sd = O.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    bs = int(g["batch"])
    x = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    m = build(cfg, sd)
    with torch.no_grad():
        y = m(x.cuda()).cpu()
    assert list(y.shape) == list(g["shape"])
    probe = y.reshape(-1)[torch.from_numpy(g["probe_index"])]
    assert float((probe - torch.from_numpy(g["probe_value"])).abs().max()) <= 1.2e-2 * 2.1      # reference probes
    with torch.no_grad():
        ref = O.swinir_forward(sd, cfg, x)
    assert float((y - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())
    assert mutual_psnr(y, ref) >= 50.0
    # BASELINE quality gate on the synthetic target of SURVEY 8(d)
    _, hr = O.synthetic_batch(bs, hw, cfg.upscale, seed=0)
    p_hip = float(O.batch_psnr(y, hr).mean())
    p_ref = float(O.batch_psnr(ref, hr).mean())
    assert abs(p_hip - p_ref) <= 0.01, f"PSNR delta {p_hip - p_ref:+.4f} dB"


@pytest.mark.parametrize("tag", ["ps4", "psd2"])
def test_tiny_gradients_vs_reference_golden(tag):
    g, cfg, sd = tiny_weights(tag)
    m = build(cfg, sd, train=True)
    x, t = torch.from_numpy(g["train.x"]).cuda(), torch.from_numpy(g["train.target"]).cuda()
    out = m(x)
    loss = torch.nn.functional.l1_loss(out, t)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["train.loss"])) <= 2e-3 * float(g["train.loss"])
    rels = []
    for n, p in m.named_parameters():
        ref = torch.from_numpy(g["grad." + n])
        assert p.grad is not None and p.grad.shape == ref.shape, n
        rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
        assert rel <= 0.1, f"{n}: relative L2 error {rel:.3e}"
        rels.append(rel)
    assert float(np.median(rels)) <= 0.04
    # accumulation semantics: a second backward doubles the gradients
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    torch.nn.functional.l1_loss(m(x), t).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-3, atol=1e-6 * float(g1[n].abs().max())), n


def test_drop_path_factors_as_data():
    """Train-mode DropPath with the per-sample keep factors passed in as data."""
    g, cfg, sd = tiny_weights("ps4")
    m = build(cfg, sd, train=True, drop_path_rate=0.3)
    x = torch.from_numpy(g["train.x"])
    gen = torch.Generator().manual_seed(3)
    keep = 1.0 - torch.linspace(0, 0.3, 4).view(4, 1, 1)
    ds = ((torch.rand(4, 2, 2, generator=gen) < keep).float() / keep)
    ds[1, 0, 0] = 0.0                                  # make sure a dropped branch is exercised
    with torch.no_grad():
        ref = O.swinir_forward(sd, cfg, x, drop_keep=ds)
        y = m(x.cuda(), drop_scale=ds.cuda()).cpu()
    assert float((y - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())
    # and the gradients under the same factors
    tgt = torch.from_numpy(g["train.target"])
    _, _, grads = O.loss_and_grads(sd, cfg, x, tgt, drop_keep=ds)
    torch.nn.functional.l1_loss(m(x.cuda(), drop_scale=ds.cuda()), tgt.cuda()).backward()
    rels = [float((p.grad.cpu() - grads[n]).norm() / (grads[n].norm() + 1e-12)) for n, p in m.named_parameters()]
    assert max(rels) <= 0.12 and float(np.median(rels)) <= 0.04
    # without explicit factors the module draws its own (eval: none)
    y1 = m(x.cuda())
    assert y1.shape == y.shape
    m.eval()
    with torch.no_grad():
        y2 = m(x.cuda()).cpu()
        assert float((y2 - O.swinir_forward(sd, cfg, x)).abs().max()) <= 1.2e-2 * float(ref.abs().max())


def test_fused_adamw_matches_oracle_update():
    from tpu_superresolution_amd._lib import check, lib
    torch.manual_seed(0)
    n = 10000
    p, g = torch.randn(n), torch.randn(n) * 3
    m, v = torch.randn(n) * 0.1, torch.rand(n) * 0.1
    pr, mr, vr = p.clone(), m.clone(), v.clone()
    total, coef = O.clip_grad_norm([g / 2.0], 1.0)     # grads averaged over world=2, then clipped to 1.0
    O.adamw_update(pr, g / 2.0 * coef, mr, vr, step=3, lr=2e-3, wd=0.01)
    pd, gd, md, vd = p.cuda(), g.cuda(), m.cuda(), v.cuda()
    ss = torch.zeros(1, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    check(lib().srk_grad_sumsq(gd.data_ptr(), n, ss.data_ptr(), st))
    check(lib().srk_adamw_clip_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, ss.data_ptr(), 1.0, 2.0, 2e-3,
                                    0.9, 0.999, 1e-8, 0.01, 3, None, st))
    assert abs(float(ss.sqrt()) / 2.0 - float(total)) <= 1e-4 * float(total)
    assert (pd.cpu() - pr).abs().max() < 2e-6 and (md.cpu() - mr).abs().max() < 1e-6 and (vd.cpu() - vr).abs().max() < 1e-6
    # a non-zero non-finite counter, or a NaN gradient norm, makes the step a no-op (weights survive for the reference's raise)
    before = (pd.clone(), md.clone(), vd.clone())
    bad = torch.ones(1, dtype=torch.int32, device="cuda")
    check(lib().srk_adamw_clip_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, ss.data_ptr(), 1.0, 2.0, 2e-3,
                                    0.9, 0.999, 1e-8, 0.01, 4, bad.data_ptr(), st))
    nan_ss = torch.full((1,), float("nan"), device="cuda")
    check(lib().srk_adamw_clip_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), n, nan_ss.data_ptr(), 1.0, 2.0,
                                    2e-3, 0.9, 0.999, 1e-8, 0.01, 4, None, st))
    assert torch.equal(pd, before[0]) and torch.equal(md, before[1]) and torch.equal(vd, before[2])


def test_train_step_fused_optimizer_moves_like_reference():
    """One full step (forward, L1, backward, clip 1.0, AdamW) vs the reference's post-step weights (G9)."""
    from tpu_superresolution_amd.optim import FusedAdamW
    g, cfg, sd = tiny_weights("psd2")
    m = build(cfg, sd, train=True)
    opt = FusedAdamW(m, lr=2e-3, weight_decay=0.01, max_grad_norm=1.0)
    x, t = torch.from_numpy(g["train.x"]).cuda(), torch.from_numpy(g["train.target"]).cuda()
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.l1_loss(m(x), t).backward()
    gn = float(opt.grad_norm())
    assert abs(gn - float(g["train.grad_norm"])) <= 3e-2 * float(g["train.grad_norm"])
    opt.step()
    agree = []
    for n, p in m.named_parameters():
        ref_post, pre = torch.from_numpy(g["post." + n]), sd[n]
        # first Adam step moves every weight by ~lr*sign(g): the step direction must agree wherever the
        # reference gradient is not in the bf16 noise floor
        d_ref, d_hip = (ref_post - pre), (p.detach().cpu() - pre)
        assert float((d_hip - d_ref).abs().max()) <= 2.05 * 2e-3 + 1e-6, n
        gref = torch.from_numpy(g["grad." + n])
        big = gref.abs() > 0.2 * gref.abs().max()
        if big.any():
            agree.append(float((torch.sign(d_ref[big]) == torch.sign(d_hip[big])).float().mean()))
    assert np.mean(agree) > 0.99
    # the next forward sees the stepped weights (re-pack) and the loss goes down on the same batch
    with torch.no_grad():
        l1 = float(torch.nn.functional.l1_loss(m(x), t))
    assert l1 < float(g["train.loss"])


def test_errors_are_loud():
    import tpu_superresolution_amd as T
    cfg = VARIANTS["ps4"]
    m = T.SwinIR(**cfg.kwargs())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 16, 16))
    with pytest.raises(ValueError, match="scale 5 is not supported"):
        T.SwinIR(**{**cfg.kwargs(), "upscale": 5})
    for bad in (dict(window_size=7, img_size=14), dict(patch_norm=False)):
        mm = T.SwinIR(**{**cfg.kwargs(), **bad}).cuda()
        with pytest.raises(NotImplementedError):
            mm(torch.rand(1, 3, 16, 16, device="cuda"))


def test_two_live_forwards_raise_instead_of_wrong_gradients():
    """One activation workspace per engine: backward of an older forward after a newer grad-enabled forward must raise
    (it would otherwise run on the newer forward's activations and return wrong gradients silently)."""
    g, cfg, sd = tiny_weights("ps4")
    m = build(cfg, sd, train=True)
    x1 = torch.from_numpy(g["x_16x16"]).cuda()
    x2 = (x1 * 0.5 + 0.1).contiguous()
    y1 = m(x1)
    y2 = m(x2)
    with pytest.raises(RuntimeError, match="overwritten by a later"):
        (y1.sum() + y2.sum()).backward()
    # the normal order still works
    for p in m.parameters():
        p.grad = None
    m(x1).sum().backward()
    assert all(p.grad is not None for p in m.parameters())


def test_eval_after_torch_optimizer_step_uses_fresh_weights():
    """train forward/backward, torch.optim.AdamW step (in-place on the flat views), then eval under no_grad: the bf16
    weight pack must be rebuilt (it used to be one step stale)."""
    g, cfg, sd = tiny_weights("ps4")
    x = torch.from_numpy(g["x_16x16"]).cuda()
    m = build(cfg, sd, train=True)
    opt = torch.optim.AdamW(m.parameters(), lr=5e-2)
    with torch.no_grad():
        m.eval()
        y0 = m(x).clone()
        m.train()
    m(x).abs().mean().backward()
    opt.step()
    m.eval()
    with torch.no_grad():
        y_auto = m(x).clone()
        m.mark_params_dirty()
        y_forced = m(x).clone()
    assert torch.equal(y_auto, y_forced)
    assert float((y_auto - y0).abs().max()) > 1e-4       # the step really moved the output
    # the same without any mode switch (in-place edits in eval mode bump the parameters' version counters)
    with torch.no_grad():
        opt.step()
        y_auto = m(x).clone()
        m.mark_params_dirty()
        assert torch.equal(y_auto, m(x))


def test_nan_batch_leaves_weights_intact():
    """finetune_swinir.py:159-165 raises on a non-finite output before backward/step; the fused step here is gated on
    the device-side counter, so the weights and Adam moments survive the bad batch."""
    from tpu_superresolution_amd.optim import FusedAdamW
    from tpu_superresolution_amd.training import assert_finite_step, train_step
    g, cfg, sd = tiny_weights("ps4")
    m = build(cfg, sd, train=True)
    opt = FusedAdamW(m, lr=1e-2, weight_decay=0.0, max_grad_norm=1.0)
    x = torch.from_numpy(g["x_16x16"]).cuda()
    hr = torch.rand(x.shape[0], 3, 64, 64, device="cuda")
    train_step(m, opt, x, hr)                                   # a good step first (moments non-zero)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    mom = (opt._m.clone(), opt._v.clone())
    xb = x.clone()
    xb[0, 0, 0, 0] = float("nan")
    loss, bad = train_step(m, opt, xb, hr)
    with pytest.raises(RuntimeError, match="non-finite"):
        assert_finite_step(loss, bad)
    after = m.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    assert torch.equal(mom[0], opt._m) and torch.equal(mom[1], opt._v)
    assert all(bool(torch.isfinite(v).all()) for v in after.values() if v.dtype.is_floating_point)


@pytest.mark.parametrize("tag", ["nc4", "dn1", "ps2_3conv_gray"])
def test_other_heads_and_3conv_gradients_vs_oracle(tag):
    """'nearest+conv' (x4: two nearest-2x stages + conv_hr), the denoising head and the '3conv' residual connection
    (network_swinir.py:466-471, :750-762): loss and every parameter gradient against the fp32 oracle, which
    tests/test_oracle_golden.py pins against the reference's own outputs for these three variants (G8)."""
    g, cfg, sd = tiny_weights(tag)
    x = torch.from_numpy(g["x_16x16"])
    t = torch.rand(x.shape[0], cfg.in_chans, 16 * cfg.upscale, 16 * cfg.upscale, generator=torch.Generator().manual_seed(5))
    loss_r, _, grads = O.loss_and_grads(sd, cfg, x, t)
    m = build(cfg, sd, train=True)
    assert [n for n, _ in m.named_parameters()] == list(grads)
    loss = torch.nn.functional.l1_loss(m(x.cuda()), t.cuda())
    loss.backward()
    # measured round 2: loss 2.0e-3 (nc4: four more bf16 activations at 16x the pixels), worst tensor 0.10 (a bias table of
    # the 3conv variant); the same runs agree with the bf16 emulation to 6e-4 / 4e-3 median (test_gpu_emulation.py)
    assert abs(float(loss.detach()) - float(loss_r)) <= 4e-3 * float(loss_r)
    rels = {n: float((p.grad.cpu() - grads[n]).norm() / (grads[n].norm() + 1e-12)) for n, p in m.named_parameters()}
    worst = max(rels, key=rels.get)
    print(f"{tag}: grad rel-L2 median {np.median(list(rels.values())):.3e}, worst {rels[worst]:.3e} ({worst})")
    assert rels[worst] <= 0.15, f"{worst}: {rels[worst]:.3e}"
    assert float(np.median(list(rels.values()))) <= 0.04
    # non-multiple-of-window training input (reflect pad + crop in both directions)
    x2 = torch.from_numpy(g["x_13x19"])
    t2 = torch.rand(x2.shape[0], cfg.in_chans, 13 * cfg.upscale, 19 * cfg.upscale, generator=torch.Generator().manual_seed(6))
    _, _, grads2 = O.loss_and_grads(sd, cfg, x2, t2)
    for p in m.parameters():
        p.grad = None
    torch.nn.functional.l1_loss(m(x2.cuda()), t2.cuda()).backward()
    rels2 = [float((p.grad.cpu() - grads2[n]).norm() / (grads2[n].norm() + 1e-12)) for n, p in m.named_parameters()]
    assert max(rels2) <= 0.15 and float(np.median(rels2)) <= 0.04


@pytest.mark.parametrize("tag", ["ps4", "ps2_3conv_gray"])
def test_forward_features_is_callable_and_matches_the_oracle(tag):
    """SwinIR.forward_features (network_swinir.py:790-803) as a method of its own: conv_first output in, normed body output
    out (fp32 NCHW), against the oracle's restatement of the same function."""
    g, cfg, sd = tiny_weights(tag)
    m = build(cfg, sd)
    f = torch.randn(2, cfg.embed_dim, 16, 24, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = O.forward_features(f, sd, cfg)
        y = m.forward_features(f.cuda()).cpu()
    assert y.shape == ref.shape == f.shape and y.dtype == torch.float32
    assert float((y - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())
    with pytest.raises(RuntimeError, match="multiples of the window size"):
        m.forward_features(torch.randn(1, cfg.embed_dim, 12, 16, device="cuda"))
    m.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        m.forward_features(f.cuda())


def test_default_init_cfg3_meets_the_survey_tolerance():
    """SURVEY 8c asks bf16-vs-fp32 max abs <= 5e-3 and mutual PSNR >= 60 dB on [0,1] images.  With weights at the reference's
    init scale (N(0, 0.02) linears; random_state_dict(scale=1.0)) the cfg3 output stays in the image range and the HIP
    path meets it (measured round 2: 1.6e-3 / 71 dB).  The other model tests use inflated weights (scale 1.5 - 3: outputs
    of range 2 - 60), hence their tolerance relative to max|ref|."""
    cfg = O.SwinIRConfig.classical_x4()
    sd = O.random_state_dict(cfg, seed=42, scale=1.0)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        y = build(cfg, sd)(x.cuda()).cpu()
        ref = O.swinir_forward(sd, cfg, x)
    assert float(ref.abs().max()) <= 1.0
    assert float((y - ref).abs().max()) <= 5e-3
    assert mutual_psnr(y, ref) >= 60.0


def test_light_whole_block_kernel_matches_layer_per_launch_path():
    """csrc/block_light.hip (one kernel per Swin block at the light width, inference) against the layer-per-launch path: the same
    rounding points, another summation order in the proj GEMM (compact K = 96 instead of the head-padded 192) -> equal up to the
    bf16 flips that causes; shifted and unshifted blocks, masked border windows, a non-multiple-of-8 input (reflect pad)."""
    from tpu_superresolution_amd._lib import check, lib
    cfg = O.SwinIRConfig.light_x2()
    sd = O.random_state_dict(cfg, seed=3, scale=1.0)
    outs = {}
    try:
        for on in (1, 0):
            check(lib().srk_set_option(b"block_light", on))
            m = build(cfg, sd)
            outs[on] = []
            for shape in ((4, 3, 48, 48), (2, 3, 40, 52)):
                x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2])).cuda()
                with torch.no_grad():
                    outs[on].append(m(x).float().cpu())
    finally:
        check(lib().srk_set_option(b"block_light", 1))
    for a, b in zip(outs[1], outs[0]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 4e-3 * float(b.abs().max())
    x = torch.rand(2, 3, 48, 48, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = O.swinir_forward(sd, cfg, x)
        got = build(cfg, sd)(x.cuda()).float().cpu()
    assert float((got - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())


def test_upsample_one_step_backward_at_embed_180_vs_oracle():
    """UpsampleOneStep (network_swinir.py:594-615, 'pixelshuffledirect') at the classical width: its conv has 12 output channels from
    180 inputs -- the small-Cout dgrad stages 83 KB of weights in LDS (it used to refuse anything above the 64 KB default).  Forward,
    loss and gradients against the CPU oracle (pinned for this head by the psd2 goldens)."""
    cfg = O.SwinIRConfig(upscale=2, in_chans=3, img_size=16, window_size=8, img_range=1.0, depths=(2,), embed_dim=180, num_heads=(6,),
                         mlp_ratio=2.0, upsampler="pixelshuffledirect", resi_connection="1conv")
    sd = O.random_state_dict(cfg, seed=9, scale=1.5)
    x, t = O.synthetic_batch(2, 16, 2, seed=10)
    m = build(cfg, sd, train=True)
    loss = torch.nn.functional.l1_loss(m(x.cuda()), t.cuda())
    loss.backward()
    ref_loss, _, grads = O.loss_and_grads(sd, cfg, x, t)
    assert abs(float(loss) - float(ref_loss)) <= 2e-3 * float(ref_loss)
    rels = {n: float((p.grad.cpu() - grads[n]).norm() / (grads[n].norm() + 1e-12)) for n, p in m.named_parameters()}
    worst = max(rels, key=rels.get)
    assert rels[worst] <= 0.15, (worst, rels[worst])
    assert float(np.median(list(rels.values()))) <= 0.04


def test_ape_forward_and_gradients_vs_reference_golden():
    """ape=True (network_swinir.py:678-689, :793-795): G15 from the imported reference -- forward at img_size x img_size, loss and every
    gradient incl. absolute_pos_embed's; any other input size fails as it does in the reference (the embedding has img_size^2 rows)."""
    import tpu_superresolution_amd as T
    from conftest import load_golden
    from test_oracle_golden import TINY_APE
    g = load_golden("g15_tiny_ape")
    cfg = O.SwinIRConfig(**TINY_APE)
    sd = O.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    m = build(cfg, sd)
    assert [n for n, _ in m.named_parameters()] == [str(s_) for s_ in g["param_order"]]
    with torch.no_grad():
        y = m(torch.from_numpy(g["x_16x16"]).cuda()).cpu()
    ref = torch.from_numpy(g["y_16x16"])
    assert float((y - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())
    m = build(cfg, sd, train=True)
    loss = torch.nn.functional.l1_loss(m(torch.from_numpy(g["train.x"]).cuda()), torch.from_numpy(g["train.target"]).cuda())
    loss.backward()
    assert abs(float(loss) - float(g["train.loss"])) <= 2e-3 * float(g["train.loss"])
    rels = {}
    for n, p_ in m.named_parameters():
        r = torch.from_numpy(g["grad." + n])
        rels[n] = float((p_.grad.cpu() - r).norm() / (r.norm() + 1e-12))
    assert max(rels.values()) <= 0.1, max(rels, key=rels.get)
    assert float(np.median(list(rels.values()))) <= 0.04        # (absolute_pos_embed itself: 0.06, inside the per-tensor bound above)
    from tpu_superresolution_amd._lib import SrkError
    with pytest.raises((SrkError, ValueError, RuntimeError), match="must match the size"):
        build(cfg, sd)(torch.rand(1, 3, 24, 24, device="cuda"))


@pytest.mark.parametrize("tag", ["ps", "psd"])
def test_window_size_16_inference_vs_reference_golden(tag):
    """G16: SwinIR(window_size=16) -- 256-token windows through the HAT path's attention kernel (swinir_w16.py): the training
    resolution, a larger size (shift masks for the actual map) and a size that needs reflect padding to a multiple of 16."""
    from test_oracle_golden import w16_weights
    from tpu_superresolution_amd._lib import SrkUnsupported
    g, cfg, sd = w16_weights(tag)
    m = build(cfg, sd)
    for hw in ((32, 32), (48, 64), (40, 24)):
        x = torch.from_numpy(g[f"{tag}.x_{hw[0]}x{hw[1]}"]).cuda()
        with torch.no_grad():
            y = m(x).cpu()
        ref = torch.from_numpy(g[f"{tag}.y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape
        err = float((y - ref).abs().max())
        print(f"w16 {tag} {hw}: max err {err:.3e} (|ref| max {float(ref.abs().max()):.3f})")
        assert err <= 2e-2 * max(1.0, float(ref.abs().max())), (tag, hw)
    with pytest.raises(SrkUnsupported, match="inference-only"):
        m.train()(torch.rand(1, 3, 32, 32, device="cuda"))


def test_window_size_16_at_width_180_runs_the_fused_kernels():
    """embed 180 / 6 heads (the classical width) with 16 x 16 windows, 64 x 64 input, batch 8: the persistent GEMMs and the fused
    MLP are the ones that run (T = 32768 tokens); checked against the oracle."""
    cfg = O.SwinIRConfig(upscale=2, in_chans=3, img_size=64, window_size=16, img_range=1.0, depths=(2,), embed_dim=180, num_heads=(6,),
                         mlp_ratio=2, upsampler="pixelshuffledirect", resi_connection="1conv")
    sd = O.random_state_dict(cfg, seed=5, scale=1.0)
    m = build(cfg, sd)
    x = torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
        want = O.swinir_forward(sd, cfg, x[:2])
    assert float((y[:2] - want).abs().max()) <= 2e-2 * max(1.0, float(want.abs().max()))
