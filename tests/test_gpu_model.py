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


@pytest.mark.parametrize("tag", ["ps4", "psd2", "ps3"])
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
        assert err <= 1.2e-2 * float(ref.abs().max()), f"{tag} {hw}: max err {err:.3e} vs ref max {float(ref.abs().max()):.3e}"
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
                                    0.9, 0.999, 1e-8, 0.01, 3, st))
    assert abs(float(ss.sqrt()) / 2.0 - float(total)) <= 1e-4 * float(total)
    assert (pd.cpu() - pr).abs().max() < 2e-6 and (md.cpu() - mr).abs().max() < 1e-6 and (vd.cpu() - vr).abs().max() < 1e-6


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
    for bad in (dict(upsampler="nearest+conv"), dict(resi_connection="3conv"), dict(window_size=7, img_size=14), dict(ape=True)):
        mm = T.SwinIR(**{**cfg.kwargs(), **bad}).cuda()
        with pytest.raises(NotImplementedError):
            mm(torch.rand(1, 3, 16, 16, device="cuda"))
