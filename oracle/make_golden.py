"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

TEST INFRASTRUCTURE.  Usage:  python -m oracle.make_golden   (from the repo root)

Every fixture is *data*: seeded inputs, weights (as arrays keyed by the reference's state_dict
names) and the outputs the reference produced for them.  No reference source text is stored.
Big configurations (BASELINE cfg2 / cfg3) store only probes / moments / SHA-1 of the fp32 output.
Weights for every case come from ``oracle.swinir_oracle.random_state_dict`` (deterministic CPU
generator), loaded into the reference model with ``strict=True`` -- which also pins the
state_dict schema.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import swinir_oracle as O  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def sha1(a: np.ndarray) -> str:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def sd_to_np(sd):
    return {"sd." + k: v.detach().cpu().numpy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def build_ref_model(ns, cfg: O.SwinIRConfig, sd):
    torch.manual_seed(0)
    m = ns.SwinIR(drop_path_rate=0.0, **cfg.kwargs())
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return m.eval()


def schema_digest(sd) -> str:
    return hashlib.sha1("\n".join(f"{k}:{tuple(v.shape)}:{v.dtype}" for k, v in sd.items()).encode()).hexdigest()


def gen_g11():
    """G11: MS_ResUNet (ms_resunet.py imports directly: torch + numpy only).  Weights from oracle.cfg1_weights (seed only is
    stored); eval forward on [1,1,128,128] (BASELINE cfg1) and on an odd size (exercises _crop_like), and one train-mode
    forward/backward (BatchNorm batch statistics) on [2,1,32,32]."""
    from oracle.cfg1_weights import fill_state_dict
    ms = import_reference("ms_resunet")
    torch.manual_seed(0)
    m = ms.MS_ResUNet()
    n_params = sum(p.numel() for p in m.parameters())
    sd = fill_state_dict(m.state_dict(), seed=11)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    arrays = dict(n_params=np.array(n_params), n_keys=np.array(len(sd)), schema_sha1=np.array(schema_digest(m.state_dict())),
                  weight_seed=np.array(11))
    m.eval()
    for tag, shape, seed in (("128", (1, 1, 128, 128), 0), ("odd", (1, 1, 50, 37), 1)):
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            y = m(x).numpy()
        assert y.shape == shape
        pg = np.random.RandomState(2).randint(0, y.size, size=64)
        arrays.update({f"{tag}.input_seed": np.array(seed), f"{tag}.shape": np.array(shape), f"{tag}.probe_index": pg,
                       f"{tag}.probe_value": y.reshape(-1)[pg], f"{tag}.mean": np.array(y.mean()), f"{tag}.std": np.array(y.std()),
                       f"{tag}.sha1": np.array(sha1(y))})
        print("G11", tag, "mean", y.mean(), "std", y.std())
    m.train()
    x = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(3))
    t = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(4))
    loss = torch.nn.functional.mse_loss(m(x), t)
    loss.backward()
    names = ["conv1.weight", "layer2.0.downsample.0.weight", "layer4.2.bn3.weight", "mflow_conv_g1_pool.0.3_outvar_dimred.weight",
             "adapt_stage3_b.0.2_conv.bias", "upCT3.weight", "clf_conv2.weight"]
    named = dict(m.named_parameters())
    arrays["train.loss"] = np.array(float(loss))
    arrays["train.grad_names"] = np.array(names)
    arrays["train.grad_norms"] = np.array([float(named[n].grad.norm()) for n in names])
    arrays["train.running_mean_bn1"] = m.bn1.running_mean.detach().numpy().copy()
    print("G11 train loss", float(loss), "params", n_params, "keys", len(sd))
    save("g11_ms_resunet", **arrays)


HAT_TINY = dict(img_size=32, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=16, compress_ratio=3,
                squeeze_factor=6, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=2.0, upscale=4, img_range=1.0, upsampler="pixelshuffle")


def gen_g13():
    """G13: HAT (hat_arch.py imports with the timm stand-in; einops is installed).  Index tables bit-exact (rpi_oca keeps its
    negative entries), a tiny HAT end to end at three input sizes incl. a non-multiple of the window (reflect pad + crop +
    dynamic mask), OCAB / HAB in isolation, and probes of the full HAT-SRx4 (BASELINE cfg4) forward."""
    from oracle import hat_oracle as HO
    ha = import_reference("hat_arch")
    # index tables
    arrays = {}
    for ws in (8, 16):
        cfg = HO.HATConfig(**{**HAT_TINY, "window_size": ws})
        m = ha.HAT(**cfg.kwargs())
        arrays[f"rpi_sa_ws{ws}"] = m.relative_position_index_SA.numpy()
        arrays[f"rpi_oca_ws{ws}"] = m.relative_position_index_OCA.numpy()
        arrays[f"mask_ws{ws}_48x32"] = (m.calculate_mask((48, 32)).numpy() != 0).astype(np.uint8)
    save("g13_hat_index", **arrays)
    # tiny end to end
    cfg = HO.HATConfig(**HAT_TINY)
    sd = HO.random_state_dict(cfg, seed=13, scale=2.0)
    torch.manual_seed(0)
    m = ha.HAT(**cfg.kwargs())
    assert list(m.state_dict().keys()) == list(sd.keys()), "schema order drifted"
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m.eval()
    arrays = dict(weight_seed=np.array(13), weight_scale=np.array(2.0),
                  weight_sha1=np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))))
    for hw in ((32, 32), (32, 48), (20, 37)):
        x = torch.rand(1, 3, *hw, generator=torch.Generator().manual_seed(hw[0] * 100 + hw[1]))
        with torch.no_grad():
            y = m(x)
        arrays[f"x_{hw[0]}x{hw[1]}"] = x.numpy()
        arrays[f"y_{hw[0]}x{hw[1]}"] = y.numpy()
    # one HAB (shifted) and the OCAB in isolation on a 32x48 token map
    xt = torch.randn(1, 32 * 48, 24, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        params = {"attn_mask": m.calculate_mask((32, 48)), "rpi_sa": m.relative_position_index_SA, "rpi_oca": m.relative_position_index_OCA}
        blk = m.layers[0].residual_group.blocks[1]
        arrays["blk.x"] = xt.numpy()
        arrays["blk.hab_shifted"] = blk(xt, (32, 48), params["rpi_sa"], params["attn_mask"]).numpy()
        arrays["blk.hab_plain"] = m.layers[0].residual_group.blocks[0](xt, (32, 48), params["rpi_sa"], params["attn_mask"]).numpy()
        arrays["blk.ocab"] = m.layers[1].residual_group.overlap_attn(xt, (32, 48), params["rpi_oca"]).numpy()
        f = torch.randn(1, 24, 32, 32, generator=torch.Generator().manual_seed(6))
        arrays["ff.x"] = f.numpy()
        arrays["ff.y"] = m.forward_features(f).numpy()
    # training-mode loss + gradients (drop_path 0): pins the oracle's autograd against the reference's
    mt = ha.HAT(drop_path_rate=0.0, **cfg.kwargs())
    mt.load_state_dict(sd, strict=True)
    mt.train()
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(7))
    t = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(8))
    loss = torch.nn.functional.l1_loss(mt(x), t)
    loss.backward()
    arrays["train.x_seed"], arrays["train.target_seed"], arrays["train.loss"] = np.array(7), np.array(8), np.array(float(loss.detach()))
    names = [n for n, _ in mt.named_parameters()]
    arrays["train.grad_norms"] = np.array([float(p.grad.norm()) for _, p in mt.named_parameters()])
    arrays["train.grad_names"] = np.array(names)
    for n, p in mt.named_parameters():
        if "layers.1.residual_group.overlap_attn" in n or "blocks.1.conv_block" in n or n.startswith(("conv_first", "conv_last")):
            arrays["grad." + n] = p.grad.numpy().copy()
    save("g13_hat_tiny", **arrays)
    # full size probe
    cfg = HO.HATConfig.sr_x4()
    sd = HO.random_state_dict(cfg, seed=42, scale=1.0)
    m = ha.HAT(**cfg.kwargs())
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    m.eval()
    xin = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        yn = m(xin).numpy()
    pg = np.random.RandomState(1).randint(0, yn.size, size=64)
    n_params = sum(p.numel() for p in m.parameters())
    save("g13_hat_cfg4_probe", probe_index=pg, probe_value=yn.reshape(-1)[pg], mean=np.array(yn.mean()), std=np.array(yn.std()),
         sha1=np.array(sha1(yn)), shape=np.array(yn.shape), n_params=np.array(n_params), n_keys=np.array(len(m.state_dict())),
         weight_seed=np.array(42), weight_scale=np.array(1.0), input_seed=np.array(0), batch=np.array(1))
    print("G13 cfg4 params", n_params, "keys", len(m.state_dict()), "mean", yn.mean(), "std", yn.std(), "max", np.abs(yn).max())


DAT_TINY = dict(img_size=32, in_chans=3, embed_dim=48, split_size=(8, 32), depth=(3, 2), num_heads=(4, 4), expansion_factor=2.0,
                upscale=2, img_range=1.0, resi_connection="1conv", upsampler="pixelshuffle")


DAT_TINY_816 = dict(img_size=32, in_chans=3, embed_dim=48, split_size=(8, 16), depth=(3, 2), num_heads=(4, 4), expansion_factor=2.0,
                    upscale=2, img_range=1.0, resi_connection="1conv", upsampler="pixelshuffle")


TINY_APE = dict(img_size=16, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=8, mlp_ratio=2, img_range=1.0,
                resi_connection="1conv", upscale=2, upsampler="pixelshuffle", ape=True)


def gen_g15():
    """G15: SwinIR with ape=True (absolute_pos_embed added after patch_embed, network_swinir.py:678-689, :793-795).  The embedding has
    img_size^2 rows, so the reference only runs at exactly img_size x img_size: forward there, and one training record (loss +
    every gradient, incl. the embedding's)."""
    ns = import_reference("network_swinir")
    cfg = O.SwinIRConfig(**TINY_APE)
    sd = O.random_state_dict(cfg, seed=16, scale=3.0)
    m = build_ref_model(ns, cfg, sd)
    assert list(m.state_dict().keys()) == [k for k, _, _ in O.state_dict_schema(cfg)]
    arrays = {"weight_seed": np.array(16), "weight_scale": np.array(3.0),
              "weight_sha1": np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()])))}
    gi = torch.Generator().manual_seed(160)
    xin = torch.rand(2, 3, 16, 16, generator=gi)
    with torch.no_grad():
        arrays["x_16x16"], arrays["y_16x16"] = xin.numpy(), m(xin).numpy()
    mt = build_ref_model(ns, cfg, sd).train()
    xt = torch.rand(2, 3, 16, 16, generator=gi)
    tgt = torch.rand(2, 3, 32, 32, generator=gi)
    loss = torch.nn.functional.l1_loss(mt(xt), tgt)
    loss.backward()
    arrays["train.x"], arrays["train.target"], arrays["train.loss"] = xt.numpy(), tgt.numpy(), loss.detach().numpy()
    for n_, p in mt.named_parameters():
        arrays["grad." + n_] = p.grad.detach().numpy().copy()
    arrays["param_order"] = np.array([n_ for n_, _ in mt.named_parameters()])
    save("g15_tiny_ape", **arrays)


def gen_g14b():
    """G14b: DAT with the reference's zero padding of q / k / v (input sizes that are not multiples of the larger split:
    dat_arch.py:376-384 + the on-the-fly masks of :404-407) and with 128-token windows -- split_size [8, 16], expansion_factor 2,
    the configuration the reference's own __main__ instantiates (dat_arch.py:862-883)."""
    from oracle import dat_oracle as DO
    da = import_reference("dat_arch")
    arrays = {}
    for tag, base in (("s832", DAT_TINY), ("s816", DAT_TINY_816)):
        cfg = DO.DATConfig(**base)
        sd = DO.random_state_dict(cfg, seed=15, scale=2.0)
        torch.manual_seed(0)
        m = da.DAT(**cfg.kwargs())
        assert list(m.state_dict().keys()) == list(sd.keys())
        for k in sd:
            if k.endswith(("rpe_biases", "relative_position_index", "attn_mask_0", "attn_mask_1")):
                assert torch.equal(m.state_dict()[k].to(sd[k].dtype), sd[k]), k
        m.load_state_dict(sd, strict=True)
        m.eval()
        arrays[f"{tag}.weight_sha1"] = np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()])))
        sizes = ((24, 40), (48, 64), (32, 32)) if tag == "s832" else ((32, 32), (24, 40), (40, 16))
        for hw in sizes:
            x = torch.rand(1, 3, *hw, generator=torch.Generator().manual_seed(hw[0] * 100 + hw[1] + 7))
            with torch.no_grad():
                y = m(x)
            arrays[f"{tag}.x_{hw[0]}x{hw[1]}"] = x.numpy()
            arrays[f"{tag}.y_{hw[0]}x{hw[1]}"] = y.numpy()
        xt = torch.randn(1, 24 * 40, 48, generator=torch.Generator().manual_seed(9))
        with torch.no_grad():
            arrays[f"{tag}.blk_shifted_24x40"] = m.layers[0].blocks[2](xt, (24, 40)).numpy()      # shifted spatial block on a padded frame
            arrays[f"{tag}.blk_plain_24x40"] = m.layers[0].blocks[0](xt, (24, 40)).numpy()
    arrays["weight_seed"], arrays["weight_scale"], arrays["blk_seed"] = np.array(15), np.array(2.0), np.array(9)
    save("g14b_dat_pad_split", **arrays)


TINY_W16 = dict(img_size=32, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=16, mlp_ratio=2, img_range=1.0,
                resi_connection="1conv", upscale=2, upsampler="pixelshuffle")


def gen_g16():
    """G16: SwinIR with window_size 16 (network_swinir.py builds any window size; 256-token windows, 961-row bias tables): the
    classical and the light-weight head, at the training resolution, at a larger size (masks recomputed, :253-257) and at a size
    that needs the reflect padding of check_image_size (:783-788)."""
    ns = import_reference("network_swinir")
    arrays = {}
    for tag, ups in (("ps", "pixelshuffle"), ("psd", "pixelshuffledirect")):
        cfg = O.SwinIRConfig(**dict(TINY_W16, upsampler=ups))
        sd = O.random_state_dict(cfg, seed=17, scale=3.0)
        m = build_ref_model(ns, cfg, sd)
        assert list(m.state_dict().keys()) == [k for k, _, _ in O.state_dict_schema(cfg)]
        arrays[f"{tag}.weight_sha1"] = np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()])))
        for hw in ((32, 32), (48, 64), (40, 24)):
            x = torch.rand(2, 3, *hw, generator=torch.Generator().manual_seed(hw[0] * 7 + hw[1]))
            with torch.no_grad():
                arrays[f"{tag}.x_{hw[0]}x{hw[1]}"], arrays[f"{tag}.y_{hw[0]}x{hw[1]}"] = x.numpy(), m(x).numpy()
    arrays["weight_seed"], arrays["weight_scale"] = np.array(17), np.array(3.0)
    save("g16_swinir_w16", **arrays)


def gen_g14c():
    """G14c: one TRAINING step of the reference's DAT (model.train(): BatchNorm with batch statistics and running-statistic updates,
    drop_path_rate 0 so that the step is deterministic): L1 loss, output, every parameter's gradient and the BatchNorm buffers after
    the step, on a 32 x 32 batch of 2 (no padding) and a 24 x 40 batch of 2 (padded window frame, on-the-fly masks)."""
    from oracle import dat_oracle as DO
    da = import_reference("dat_arch")
    arrays = {}
    cfg = DO.DATConfig(**DAT_TINY)
    sd = DO.random_state_dict(cfg, seed=16, scale=2.0)
    arrays["weight_seed"], arrays["weight_scale"] = np.array(16), np.array(2.0)
    arrays["weight_sha1"] = np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()])))
    for hw in ((32, 32), (24, 40)):
        tag = f"{hw[0]}x{hw[1]}"
        torch.manual_seed(0)
        m = da.DAT(**cfg.kwargs(), drop_path_rate=0.0)
        m.load_state_dict(sd, strict=True)
        m.train()
        g = torch.Generator().manual_seed(hw[0] * 10 + hw[1])
        x = torch.rand(2, 3, *hw, generator=g)
        t = torch.rand(2, 3, hw[0] * 2, hw[1] * 2, generator=g)
        y = m(x)
        loss = torch.nn.functional.l1_loss(y, t)
        loss.backward()
        arrays[f"{tag}.x"], arrays[f"{tag}.t"], arrays[f"{tag}.y"] = x.numpy(), t.numpy(), y.detach().numpy()
        arrays[f"{tag}.loss"] = np.array(float(loss))
        for n, p_ in m.named_parameters():
            arrays[f"{tag}.grad.{n}"] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy()
        for n, b_ in m.named_buffers():
            if n.endswith(("running_mean", "running_var", "num_batches_tracked")):
                arrays[f"{tag}.buf.{n}"] = b_.numpy()
    save("g14c_dat_train", **arrays)


def gen_g14():
    """G14: DAT (dat_arch.py imports with the timm stand-in).  Index tables bit-exact, a tiny DAT end to end (eval: BatchNorm with
    running statistics) incl. a shifted spatial block (rg 0, b 2), a channel-attention block and both window orientations, the
    blocks in isolation, and probes of the full DAT x4 (BASELINE cfg5) forward."""
    from oracle import dat_oracle as DO
    da = import_reference("dat_arch")
    cfg = DO.DATConfig(**DAT_TINY)
    sd = DO.random_state_dict(cfg, seed=14, scale=2.0)
    torch.manual_seed(0)
    m = da.DAT(**cfg.kwargs())
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), [(a, b) for a, b in zip(ref_sd.keys(), sd.keys()) if a != b][:5]
    for k in sd:      # the reference's own buffers equal the closed forms
        if k.endswith(("rpe_biases", "relative_position_index", "attn_mask_0", "attn_mask_1")):
            assert torch.equal(ref_sd[k].to(sd[k].dtype), sd[k]), k
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m.eval()
    arrays = dict(weight_seed=np.array(14), weight_scale=np.array(2.0),
                  weight_sha1=np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))))
    blk0 = m.layers[0].blocks[0].attn.attns
    arrays["rpi_8x32"] = blk0[0].relative_position_index.numpy()
    arrays["rpi_32x8"] = blk0[1].relative_position_index.numpy()
    arrays["rpe_8x32"] = blk0[0].rpe_biases.numpy()
    mk = m.layers[0].blocks[2].attn.calculate_mask(32, 64)
    arrays["mask0_32x64"] = (mk[0].numpy() != 0).astype(np.uint8)
    arrays["mask1_32x64"] = (mk[1].numpy() != 0).astype(np.uint8)
    for hw in ((32, 32), (32, 64)):
        x = torch.rand(1, 3, *hw, generator=torch.Generator().manual_seed(hw[0] * 100 + hw[1]))
        with torch.no_grad():
            y = m(x)
        arrays[f"x_{hw[0]}x{hw[1]}"] = x.numpy()
        arrays[f"y_{hw[0]}x{hw[1]}"] = y.numpy()
    xt = torch.randn(1, 32 * 64, 48, generator=torch.Generator().manual_seed(5))        # 32x64: masks computed on the fly (:404-407)
    xs = torch.randn(1, 32 * 32, 48, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        arrays["blk.x_seed"] = np.array(5)
        arrays["blk.xs_seed"] = np.array(6)
        arrays["blk.spatial_shifted"] = m.layers[0].blocks[2](xt, (32, 64)).numpy()
        arrays["blk.spatial_plain"] = m.layers[0].blocks[0](xs, (32, 32)).numpy()
        arrays["blk.channel"] = m.layers[0].blocks[1](xs, (32, 32)).numpy()
        arrays["blk.shifted_rg1"] = m.layers[1].blocks[0](xs, (32, 32)).numpy()
        arrays["blk.sgfn"] = m.layers[0].blocks[0].ffn(xs, 32, 32).numpy()
    save("g14_dat_tiny", **arrays)
    cfg = DO.DATConfig.sr_x4()
    sd = DO.random_state_dict(cfg, seed=42, scale=1.0)
    m = da.DAT(**cfg.kwargs())
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    m.eval()
    xin = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        yn = m(xin).numpy()
    pg = np.random.RandomState(1).randint(0, yn.size, size=64)
    n_params = sum(p.numel() for p in m.parameters())
    save("g14_dat_cfg5_probe", probe_index=pg, probe_value=yn.reshape(-1)[pg], mean=np.array(yn.mean()), std=np.array(yn.std()),
         sha1=np.array(sha1(yn)), shape=np.array(yn.shape), n_params=np.array(n_params), n_keys=np.array(len(m.state_dict())),
         weight_seed=np.array(42), weight_scale=np.array(1.0), input_seed=np.array(0), batch=np.array(1))
    print("G14 cfg5 params", n_params, "keys", len(m.state_dict()), "mean", yn.mean(), "std", yn.std(), "max", np.abs(yn).max())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--only-g14" in sys.argv:
        return gen_g14()
    if "--only-g11" in sys.argv:
        return gen_g11()
    if "--only-g13" in sys.argv:
        return gen_g13()
    if "--only-g14b" in sys.argv:
        return gen_g14b()
    if "--only-g14c" in sys.argv:
        return gen_g14c()
    if "--only-g16" in sys.argv:
        return gen_g16()
    if "--only-g15" in sys.argv:
        return gen_g15()
    gen_g11()
    gen_g13()
    gen_g14()
    gen_g14b()
    gen_g15()
    gen_g14c()
    gen_g16()
    ns = import_reference("network_swinir")

    # ---- G1/G2: index maps ------------------------------------------------------------------
    x = torch.arange(2 * 16 * 24 * 3, dtype=torch.int32).reshape(2, 16, 24, 3)
    wp = ns.window_partition(x, 8)
    wr = ns.window_reverse(wp, 8, 16, 24)
    assert torch.equal(wr, x)
    big = torch.arange(32 * 64 * 64 * 180, dtype=torch.int32).reshape(32, 64, 64, 180)
    bigp = ns.window_partition(big, 8)
    rolled = torch.roll(x, shifts=(-4, -4), dims=(1, 2))
    rolled_back = torch.roll(x, shifts=(4, 4), dims=(1, 2))
    rp = ns.window_partition(rolled, 8)
    save("g1_g2_index_maps", x=x.numpy(), partition=wp.numpy(), roll_m4=rolled.numpy(),
         roll_p4=rolled_back.numpy(), roll_m4_partition=rp.numpy(),
         cfg3_partition_sha1=np.array(sha1(bigp.numpy())),
         cfg3_roll_partition_sha1=np.array(sha1(ns.window_partition(torch.roll(big, (-4, -4), (1, 2)), 8).numpy())))
    del big, bigp

    # ---- G3: relative position index ----------------------------------------------------------
    rpis = {}
    for ws in (7, 8, 16):
        wa = ns.WindowAttention(12, (ws, ws), 2)
        rpis[f"rpi_ws{ws}"] = wa.relative_position_index.numpy()
    assert sha1(rpis["rpi_ws8"]).startswith("8520653ae68ea9ff"), sha1(rpis["rpi_ws8"])
    save("g3_rpi", **rpis)

    # ---- G4: shift masks ------------------------------------------------------------------------
    blk = ns.SwinTransformerBlock(12, (64, 64), 2, window_size=8, shift_size=4, mlp_ratio=2)
    masks = {}
    for hw in ((64, 64), (48, 48), (16, 24), (24, 40)):
        m = blk.calculate_mask(hw)
        vals = set(torch.unique(m).tolist())
        assert vals <= {0.0, -100.0}
        masks[f"mask_{hw[0]}x{hw[1]}"] = (m != 0).numpy().astype(np.uint8)
    assert int(masks["mask_64x64"].reshape(64, -1).any(axis=1).sum()) == 15
    save("g4_masks", **masks)

    # ---- G5: WindowAttention ------------------------------------------------------------------
    g = torch.Generator().manual_seed(5)
    arrays = {}
    for tag, dim, nH in (("a", 24, 2), ("b", 60, 6)):
        wa = ns.WindowAttention(dim, (8, 8), nH).eval()
        with torch.no_grad():
            for p in wa.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.ndim > 1 else 0.1))
        xw = torch.randn(2 * 4, 64, dim, generator=g)
        blk = ns.SwinTransformerBlock(dim, (16, 16), nH, window_size=8, shift_size=4, mlp_ratio=2)
        mask = blk.calculate_mask((16, 16))
        with torch.no_grad():
            y0 = wa(xw, None)
            y1 = wa(xw, mask)
        arrays.update({f"{tag}.x": xw.numpy(), f"{tag}.y_nomask": y0.numpy(), f"{tag}.y_mask": y1.numpy(),
                       f"{tag}.nH": np.array(nH)})
        arrays.update({f"{tag}.sd.{k}": v.numpy() for k, v in wa.state_dict().items()})
    save("g5_window_attention", **arrays)

    # ---- G6: SwinTransformerBlock ----------------------------------------------------------------
    arrays = {}
    for shift in (0, 4):
        blk = ns.SwinTransformerBlock(24, (16, 16), 2, window_size=8, shift_size=shift, mlp_ratio=2).eval()
        with torch.no_grad():
            for n_, p in blk.named_parameters():
                if "norm" in n_ and n_.endswith("weight"):
                    p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.ndim > 1 else 0.1))
        xb = torch.randn(2, 256, 24, generator=g)
        with torch.no_grad():
            yb = blk(xb, (16, 16))
            # different x_size than input_resolution -> dynamic mask path (:259-262)
            xb2 = torch.randn(1, 16 * 24, 24, generator=g)
            yb2 = blk(xb2, (16, 24))
        arrays.update({f"s{shift}.x": xb.numpy(), f"s{shift}.y": yb.numpy(),
                       f"s{shift}.x_16x24": xb2.numpy(), f"s{shift}.y_16x24": yb2.numpy()})
        arrays.update({f"s{shift}.sd.{k}": v.numpy() for k, v in blk.state_dict().items()})
    save("g6_swin_block", **arrays)

    # ---- G7: PixelShuffle / Upsample / UpsampleOneStep -----------------------------------------------
    arrays = {}
    xi = torch.arange(2 * 36 * 3 * 5, dtype=torch.int32).reshape(2, 36, 3, 5)
    arrays["ps.x"] = xi.numpy()
    arrays["ps.r2"] = torch.pixel_shuffle(xi.float(), 2).int().numpy()
    arrays["ps.r3"] = torch.pixel_shuffle(xi.float(), 3).int().numpy()
    for scale in (2, 3, 4):
        up = ns.Upsample(scale, 8).eval()
        xu = torch.randn(1, 8, 6, 5, generator=g)
        with torch.no_grad():
            yu = up(xu)
        arrays.update({f"up{scale}.x": xu.numpy(), f"up{scale}.y": yu.numpy()})
        arrays.update({f"up{scale}.sd.{k}": v.numpy() for k, v in up.state_dict().items()})
    one = ns.UpsampleOneStep(2, 12, 3).eval()
    xo = torch.randn(2, 12, 6, 5, generator=g)
    with torch.no_grad():
        yo = one(xo)
    arrays.update({"one.x": xo.numpy(), "one.y": yo.numpy()})
    arrays.update({f"one.sd.{k}": v.numpy() for k, v in one.state_dict().items()})
    try:
        ns.Upsample(5, 8)
        raise AssertionError("scale 5 must raise")
    except ValueError as e:
        arrays["up5.error"] = np.array(str(e))
    save("g7_upsample", **arrays)

    # ---- G8/G9: tiny end-to-end models + one training step ---------------------------------------------
    tiny = dict(img_size=16, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=8,
                mlp_ratio=2, img_range=1.0, resi_connection="1conv")
    variants = {
        "ps4": O.SwinIRConfig(upscale=4, upsampler="pixelshuffle", **tiny),
        "psd2": O.SwinIRConfig(upscale=2, upsampler="pixelshuffledirect", **tiny),
        "ps3": O.SwinIRConfig(upscale=3, upsampler="pixelshuffle", **tiny),
        "nc4": O.SwinIRConfig(upscale=4, upsampler="nearest+conv", **tiny),
        "dn1": O.SwinIRConfig(upscale=1, upsampler="", **tiny),
        "ps2_3conv_gray": O.SwinIRConfig(upscale=2, upsampler="pixelshuffle",
                                         **{**tiny, "in_chans": 1, "resi_connection": "3conv", "embed_dim": 32}),
    }
    for tag, cfg in variants.items():
        sd = O.random_state_dict(cfg, seed=8, scale=3.0)
        m = build_ref_model(ns, cfg, sd)
        ref_keys = list(m.state_dict().keys())
        assert ref_keys == [k for k, _, _ in O.state_dict_schema(cfg)], tag
        # weights are regenerated from (seed, scale) by the tests; only their digest is stored
        arrays = {"weight_seed": np.array(8), "weight_scale": np.array(3.0),
                  "weight_sha1": np.array(sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1)
                                                               for v in sd.values()])))}
        gi = torch.Generator().manual_seed(80)
        for hw in ((16, 16), (13, 19), (24, 32)):
            xin = torch.rand(2, cfg.in_chans, *hw, generator=gi)
            with torch.no_grad():
                y = m(xin)
            arrays[f"x_{hw[0]}x{hw[1]}"] = xin.numpy()
            arrays[f"y_{hw[0]}x{hw[1]}"] = y.numpy()
        if tag in ("ps4", "psd2"):
            # G9: L1 loss, grads, clip 1.0, one AdamW step (finetune_swinir.py:154-176), fp32 CPU
            mt = build_ref_model(ns, cfg, sd).train()          # drop_path_rate = 0 -> deterministic
            xin = torch.rand(2, 3, 16, 16, generator=gi)
            tgt = torch.rand(2, 3, 16 * cfg.upscale, 16 * cfg.upscale, generator=gi)
            opt = torch.optim.AdamW([p for p in mt.parameters() if p.requires_grad], lr=2e-3, weight_decay=0.01)
            opt.zero_grad(set_to_none=True)
            out = mt(xin)
            loss = torch.nn.functional.l1_loss(out, tgt)
            loss.backward()
            arrays["train.x"] = xin.numpy()
            arrays["train.target"] = tgt.numpy()
            arrays["train.loss"] = loss.detach().numpy()
            for n_, p in mt.named_parameters():
                arrays["grad." + n_] = p.grad.detach().numpy().copy()
            total = torch.nn.utils.clip_grad_norm_(mt.parameters(), 1.0)
            arrays["train.grad_norm"] = total.detach().numpy()
            opt.step()
            if tag == "psd2":                                   # AdamW result: the small variant is enough
                for n_, p in mt.named_parameters():
                    arrays["post." + n_] = p.detach().numpy().copy()
        save(f"g8_tiny_{tag}", **arrays)

    # ---- G10: BASELINE cfg2 / cfg3 probes ---------------------------------------------------------------------
    for tag, cfg, bs, hw, wscale in (("cfg2", O.SwinIRConfig.light_x2(), 2, 48, 1.0),
                                     ("cfg3", O.SwinIRConfig.classical_x4(), 1, 64, 1.5)):
        sd = O.random_state_dict(cfg, seed=42, scale=wscale)
        m = build_ref_model(ns, cfg, sd)
        n_params = sum(p.numel() for p in m.parameters())
        xin = torch.rand(bs, 3, hw, hw, generator=torch.Generator().manual_seed(0))
        with torch.no_grad():
            y = m(xin)
        yn = y.numpy()
        pg = np.random.RandomState(1).randint(0, yn.size, size=64)
        save(f"g10_{tag}_probe", probe_index=pg, probe_value=yn.reshape(-1)[pg], mean=np.array(yn.mean()),
             std=np.array(yn.std()), sha1=np.array(sha1(yn)), shape=np.array(yn.shape), n_params=np.array(n_params),
             n_keys=np.array(len(m.state_dict())), weight_seed=np.array(42), weight_scale=np.array(wscale),
             input_seed=np.array(0), batch=np.array(bs))
        print(tag, "params", n_params, "keys", len(m.state_dict()), "mean", yn.mean(), "std", yn.std())

    # ---- G12: PSNR formulas restated from text (finetune_swinir.py:69-74, evaluate.py:24-29) ------------------
    gp = torch.Generator().manual_seed(12)
    a = torch.rand(3, 3, 20, 20, generator=gp) * 1.2 - 0.1
    b = torch.rand(3, 3, 20, 20, generator=gp)
    pa, pb = a.clamp(0, 1), b.clamp(0, 1)
    mse = torch.nn.functional.mse_loss(pa, pb, reduction="none").view(3, -1).mean(dim=1)
    bp = 20.0 * torch.log10(1.0 / torch.sqrt(mse + 1e-8))
    mse2 = torch.clamp(torch.mean((a - b) ** 2, dim=[1, 2, 3]), min=1e-10)
    ep = float((20.0 * torch.log10(1.0 / torch.sqrt(mse2))).mean())
    save("g12_psnr", a=a.numpy(), b=b.numpy(), batch_psnr=bp.numpy(), eval_psnr=np.array(ep))


if __name__ == "__main__":
    main()
