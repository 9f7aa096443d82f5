"""HIP SwinIR path against the bf16-rounding emulation (oracle/bf16_emulation.py).

The fp32 oracle bounds the HIP path only to bf16 noise (test_gpu_model.py: 1.2 % forward, 10 % per
gradient tensor).  The emulation rounds where the HIP pipeline rounds, so the two agree up to
accumulation order and the 1-ulp bf16 flips that a different summation order causes (which then
propagate through the following layers, so the agreement is not exact).  Measured on MI355X (round 1):
forward 0.04 - 0.37 % of the output range, gradient tensors median 0.007 - 0.5 %, worst 2.7 % (a
relative-position bias table, a small-norm reduction) -- about 5x below the distance to the fp32
oracle, which is what catches logic errors (index maps, LayerNorm-backward algebra, DropPath scaling,
gradient routing) that would hide under the bf16 noise floor.

Tolerances: forward max|err| <= 6e-3 * max|emul|; loss relative error <= 1e-3;
            gradients per-tensor relative L2 <= 5e-2, median <= 1e-2.
"""
import numpy as np
import pytest
import torch

from oracle import bf16_emulation as E
from test_gpu_model import build
from test_oracle_golden import tiny_weights

pytestmark = pytest.mark.gpu

FWD_TOL, LOSS_TOL, GRAD_TOL, GRAD_MEDIAN_TOL = 6e-3, 1e-3, 5e-2, 1e-2


def _drop_factors(n_blocks, B, rate, seed):
    gen = torch.Generator().manual_seed(seed)
    keep = 1.0 - torch.linspace(0, rate, n_blocks).view(n_blocks, 1, 1)
    ds = (torch.rand(n_blocks, 2, B, generator=gen) < keep).float() / keep
    ds[1, 0, 0] = 0.0
    return ds


@pytest.mark.parametrize("tag,drop", [("ps4", False), ("psd2", False), ("ps3", False), ("ps4", True), ("nc4", False), ("dn1", False),
                                      ("ps2_3conv_gray", False), ("ps2_3conv_gray", True)])
def test_forward_and_gradients_match_bf16_emulation(tag, drop):
    g, cfg, sd = tiny_weights(tag)
    if "train.x" in g:
        x, t = torch.from_numpy(g["train.x"]), torch.from_numpy(g["train.target"])
    else:
        x = torch.from_numpy(g["x_16x16"])
        t = torch.rand(x.shape[0], x.shape[1], 16 * cfg.upscale, 16 * cfg.upscale, generator=torch.Generator().manual_seed(5))
    ds = _drop_factors(sum(cfg.depths), x.shape[0], 0.3, 3) if drop else None
    loss_e, out_e, grads_e = E.loss_and_grads_emul(sd, cfg, x, t, drop_keep=ds)
    # d(L1)/d(out) = sign(out - t) / N: a pixel whose output lands within the bf16 noise of its target can flip sign between the
    # two implementations, and ONE flip among ~1e3 outputs is a 6 % relative change of every gradient tensor.  Move such targets
    # away from the output (the comparison is about the backward algebra, not about ties of the loss).
    margin = 2.0 * (1e-2 if cfg.resi_connection == "3conv" else FWD_TOL) * float(out_e.abs().max())     # twice the forward tolerance
    close = (out_e - t).abs() < margin
    if bool(close.any()):
        t = torch.where(close, out_e + torch.where(t >= out_e, 5.0 * margin, -5.0 * margin), t)
        loss_e, out_e, grads_e = E.loss_and_grads_emul(sd, cfg, x, t, drop_keep=ds)

    m = build(cfg, sd, train=True, drop_path_rate=0.3 if drop else 0.0)
    out = m(x.cuda(), drop_scale=None if ds is None else ds.cuda())
    loss = torch.nn.functional.l1_loss(out, t.cuda())
    loss.backward()

    err = float((out.detach().cpu() - out_e).abs().max())
    # '3conv': a bf16 flip in one of the C/4 = 8 narrow channels moves an output ~4x more than a flip among 24-32 channels
    fwd_tol = 1e-2 if cfg.resi_connection == "3conv" else FWD_TOL
    assert err <= fwd_tol * float(out_e.abs().max()), f"forward max err {err:.3e} (|emul|max {float(out_e.abs().max()):.3e})"
    assert abs(float(loss) - float(loss_e)) <= LOSS_TOL * float(loss_e)
    rels = {}
    for n, p in m.named_parameters():
        ref = grads_e[n]
        rels[n] = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
    worst = max(rels, key=rels.get)
    print(f"{tag} drop={drop}: fwd err {err:.3e}, grad rel-L2 median {np.median(list(rels.values())):.3e}, worst {rels[worst]:.3e} ({worst})")
    # '3conv' (C/4 = 8 narrow channels) is ~3x noisier than the other variants, most of all in the smallest-norm tensors (the
    # bias tables, whose gradient shrinks further when DropPath removes a sample's branch): measured worst 0.14 / median 1.1e-2
    g_tol, g_med = (0.2, 2e-2) if cfg.resi_connection == "3conv" else (GRAD_TOL, GRAD_MEDIAN_TOL)
    assert rels[worst] <= g_tol, f"{worst}: relative L2 error {rels[worst]:.3e}"
    assert float(np.median(list(rels.values()))) <= g_med


def test_non_multiple_of_window_input_matches_emulation():
    """reflect-pad + crop path (network_swinir.py:827-833, :843) in inference."""
    g, cfg, sd = tiny_weights("ps4")
    m = build(cfg, sd)
    for hw in ((13, 19), (24, 32)):
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            ref = E.swinir_forward_emul(sd, cfg, x)
            y = m(x.cuda()).cpu()
        assert float((y - ref).abs().max()) <= FWD_TOL * float(ref.abs().max())


@pytest.mark.parametrize("tag", ["nc4", "dn1", "ps2_3conv_gray"])
def test_other_heads_forward_all_sizes_match_emulation(tag):
    """'nearest+conv', the denoising head and '3conv' (narrow C/4-channel intermediates: fewer terms per sum, so the
    distance to the fp32 oracle is ~1.2 % of the range for the gray 32-channel variant; against the emulation, which rounds
    at the same points, the agreement is 0.2 - 0.8 % of the range; measured round 2)."""
    g, cfg, sd = tiny_weights(tag)
    m = build(cfg, sd)
    for hw in ((16, 16), (13, 19), (24, 32)):
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            ref = E.swinir_forward_emul(sd, cfg, x)
            y = m(x.cuda()).cpu()
        err = float((y - ref).abs().max())
        print(f"{tag} {hw}: max err vs emulation {err:.3e} ({err / float(ref.abs().max()):.2e} of range)")
        assert err <= (1e-2 if cfg.resi_connection == "3conv" else FWD_TOL) * float(ref.abs().max())
