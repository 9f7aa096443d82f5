"""Pin the CPU oracle against golden vectors produced by the reference itself (CPU-only)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, sub_state_dict
from oracle import swinir_oracle as O

FP32_TOL = 1e-5


def _sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_g1_window_partition_reverse_bit_exact():
    g = load_golden("g1_g2_index_maps")
    x = g["x"]
    wp = O.np_window_partition(x, 8)
    assert wp.dtype == x.dtype and np.array_equal(wp, g["partition"])
    assert np.array_equal(O.np_window_reverse(wp, 8, 16, 24), x)


def test_g1_cfg3_size_partition_sha1():
    g = load_golden("g1_g2_index_maps")
    big = np.arange(32 * 64 * 64 * 180, dtype=np.int32).reshape(32, 64, 64, 180)
    assert _sha1(O.np_window_partition(big, 8)) == str(g["cfg3_partition_sha1"])
    idx = O.window_token_index(64, 64, 8, 4).reshape(-1)
    fused = big.reshape(32, 4096, 180)[:, idx].reshape(32 * 64, 8, 8, 180)
    assert _sha1(fused) == str(g["cfg3_roll_partition_sha1"])


def test_g2_roll_bit_exact_and_fused_gather():
    g = load_golden("g1_g2_index_maps")
    x = g["x"]
    assert np.array_equal(O.np_roll2d(x, -4, -4), g["roll_m4"])
    assert np.array_equal(O.np_roll2d(x, 4, 4), g["roll_p4"])
    idx = O.window_token_index(16, 24, 8, 4)
    fused = x.reshape(2, 16 * 24, 3)[:, idx.reshape(-1)].reshape(-1, 8, 8, 3)
    assert np.array_equal(fused, g["roll_m4_partition"])
    # the scatter with the same map inverts it (window_reverse + roll(+shift))
    back = np.empty_like(x.reshape(2, -1, 3))
    back[:, idx.reshape(-1)] = fused.reshape(2, -1, 3)
    assert np.array_equal(back.reshape(x.shape), x)


@pytest.mark.parametrize("ws", [7, 8, 16])
def test_g3_relative_position_index(ws):
    g = load_golden("g3_rpi")
    rpi = O.relative_position_index(ws)
    assert rpi.dtype == np.int64 and np.array_equal(rpi, g[f"rpi_ws{ws}"])
    if ws == 8:
        assert _sha1(rpi).startswith("8520653ae68ea9ff")


@pytest.mark.parametrize("hw", [(64, 64), (48, 48), (16, 24), (24, 40)])
def test_g4_shift_mask(hw):
    g = load_golden("g4_masks")
    m = O.shift_attn_mask(hw[0], hw[1], 8, 4)
    assert set(np.unique(m).tolist()) <= {0.0, -100.0}
    assert np.array_equal((m != 0).astype(np.uint8), g[f"mask_{hw[0]}x{hw[1]}"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g5_window_attention(tag):
    g = load_golden("g5_window_attention")
    sd = sub_state_dict(g, f"{tag}.sd.")
    x = torch.from_numpy(g[f"{tag}.x"])
    nH = int(g[f"{tag}.nH"])
    y0 = O.window_attention(x, sd, "", nH, 8, None)
    mask = torch.from_numpy(O.shift_attn_mask(16, 16, 8, 4))
    y1 = O.window_attention(x, sd, "", nH, 8, mask)
    assert (y0 - torch.from_numpy(g[f"{tag}.y_nomask"])).abs().max() < FP32_TOL
    assert (y1 - torch.from_numpy(g[f"{tag}.y_mask"])).abs().max() < FP32_TOL


@pytest.mark.parametrize("shift", [0, 4])
def test_g6_swin_block(shift):
    g = load_golden("g6_swin_block")
    sd = sub_state_dict(g, f"s{shift}.sd.")
    for suffix, hw in (("", (16, 16)), ("_16x24", (16, 24))):
        x = torch.from_numpy(g[f"s{shift}.x{suffix}"])
        y = O.swin_block(x, hw, sd, "", 2, 8, shift)
        assert (y - torch.from_numpy(g[f"s{shift}.y{suffix}"])).abs().max() < FP32_TOL


def test_g7_pixel_shuffle_and_upsample():
    g = load_golden("g7_upsample")
    x = g["ps.x"]
    assert np.array_equal(O.np_pixel_shuffle(x, 2), g["ps.r2"])
    assert np.array_equal(O.np_pixel_shuffle(x, 3), g["ps.r3"])
    assert torch.equal(O.pixel_shuffle(torch.from_numpy(x), 3), torch.from_numpy(g["ps.r3"]))
    for scale in (2, 3, 4):
        sd = sub_state_dict(g, f"up{scale}.sd.")
        f = torch.from_numpy(g[f"up{scale}.x"])
        if scale == 3:
            f = O.pixel_shuffle(torch.nn.functional.conv2d(f, sd["0.weight"], sd["0.bias"], padding=1), 3)
        else:
            for i in range(scale // 2):
                f = O.pixel_shuffle(torch.nn.functional.conv2d(f, sd[f"{2*i}.weight"], sd[f"{2*i}.bias"], padding=1), 2)
        assert (f - torch.from_numpy(g[f"up{scale}.y"])).abs().max() < FP32_TOL
    sd = sub_state_dict(g, "one.sd.")
    f = O.pixel_shuffle(torch.nn.functional.conv2d(torch.from_numpy(g["one.x"]), sd["0.weight"], sd["0.bias"], padding=1), 2)
    assert (f - torch.from_numpy(g["one.y"])).abs().max() < FP32_TOL
    assert "scale 5 is not supported" in str(g["up5.error"])


TINY = dict(img_size=16, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=8,
            mlp_ratio=2, img_range=1.0, resi_connection="1conv")
VARIANTS = {
    "ps4": O.SwinIRConfig(upscale=4, upsampler="pixelshuffle", **TINY),
    "psd2": O.SwinIRConfig(upscale=2, upsampler="pixelshuffledirect", **TINY),
    "ps3": O.SwinIRConfig(upscale=3, upsampler="pixelshuffle", **TINY),
    "nc4": O.SwinIRConfig(upscale=4, upsampler="nearest+conv", **TINY),
    "dn1": O.SwinIRConfig(upscale=1, upsampler="", **TINY),
    "ps2_3conv_gray": O.SwinIRConfig(upscale=2, upsampler="pixelshuffle",
                                     **{**TINY, "in_chans": 1, "resi_connection": "3conv", "embed_dim": 32}),
}


def tiny_weights(tag):
    g = load_golden(f"g8_tiny_{tag}")
    cfg = VARIANTS[tag]
    sd = O.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g["weight_sha1"]), "weight generator drifted from the one the fixtures were made with"
    return g, cfg, sd


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_g8_tiny_end_to_end(tag):
    g, cfg, sd = tiny_weights(tag)
    for hw in ((16, 16), (13, 19), (24, 32)):
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = O.swinir_forward(sd, cfg, x)
        ref = torch.from_numpy(g[f"y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape == (2, cfg.in_chans, hw[0] * cfg.upscale, hw[1] * cfg.upscale)
        assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("tag", ["ps4", "psd2"])
def test_g9_train_step(tag):
    g, cfg, sd = tiny_weights(tag)
    x = torch.from_numpy(g["train.x"])
    t = torch.from_numpy(g["train.target"])
    state = O.TrainState(sd={k: v.clone() for k, v in sd.items()})
    loss, total, grads = O.train_step(state, cfg, x, t, lr=2e-3, wd=0.01, grad_clip=1.0)
    assert abs(float(loss) - float(g["train.loss"])) < 1e-6
    assert abs(float(total) - float(g["train.grad_norm"])) < 1e-4 * float(g["train.grad_norm"])
    for k, gr in grads.items():
        ref = torch.from_numpy(g["grad." + k])
        assert (gr - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max())), k
    if tag == "psd2":
        for k in grads:
            ref = torch.from_numpy(g["post." + k])
            assert (state.sd[k] - ref).abs().max() < 2e-6, k


@pytest.mark.parametrize("tag,cfg,hw", [("cfg2", O.SwinIRConfig.light_x2(), 48), ("cfg3", O.SwinIRConfig.classical_x4(), 64)])
def test_g10_full_size_probes(tag, cfg, hw):
    g = load_golden(f"g10_{tag}_probe")
    sd = O.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    assert len(sd) == int(g["n_keys"])
    assert sum(v.numel() for k, v in sd.items() if k in set(O.param_keys(cfg))) == int(g["n_params"])
    x = torch.rand(int(g["batch"]), 3, hw, hw, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.no_grad():
        y = O.swinir_forward(sd, cfg, x).numpy()
    assert list(y.shape) == list(g["shape"])
    assert np.abs(y.reshape(-1)[g["probe_index"]] - g["probe_value"]).max() < 2e-5
    assert abs(float(y.mean()) - float(g["mean"])) < 1e-5 and abs(float(y.std()) - float(g["std"])) < 1e-5


def test_g12_psnr_formulas():
    g = load_golden("g12_psnr")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    assert (O.batch_psnr(a, b) - torch.from_numpy(g["batch_psnr"])).abs().max() < 1e-4
    assert abs(O.eval_psnr(a, b) - float(g["eval_psnr"])) < 1e-4


def test_unsupported_scale_raises_like_reference():
    cfg = O.SwinIRConfig(upscale=5, upsampler="pixelshuffle", **TINY)
    with pytest.raises(ValueError, match="scale 5 is not supported"):
        O.state_dict_schema(cfg)


def test_bf16_emulation_stays_within_bf16_noise_of_the_fp32_oracle():
    """oracle/bf16_emulation.py is the fp32 oracle plus rounding hooks: it must stay within bf16 noise of the
    golden fp32 outputs and gradients (it is the tighter checker of tests/test_gpu_emulation.py)."""
    from oracle import bf16_emulation as E
    g, cfg, sd = tiny_weights("ps4")
    x, t = torch.from_numpy(g["train.x"]), torch.from_numpy(g["train.target"])
    loss, out, grads = E.loss_and_grads_emul(sd, cfg, x, t)
    assert abs(float(loss) - float(g["train.loss"])) <= 2e-3 * float(g["train.loss"])
    rels = [float((grads[k] - torch.from_numpy(g["grad." + k])).norm() / (torch.from_numpy(g["grad." + k]).norm() + 1e-12)) for k in grads]
    assert max(rels) <= 0.1 and float(np.median(rels)) <= 0.04
    with torch.no_grad():
        y = E.swinir_forward_emul(sd, cfg, torch.from_numpy(g["x_13x19"]))
    ref = torch.from_numpy(g["y_13x19"])
    assert y.shape == ref.shape and float((y - ref).abs().max()) <= 1.2e-2 * float(ref.abs().max())


# ---- G13: HAT (reference modules/hat_arch.py; fixtures by oracle/make_golden.py::gen_g13) -----------------------------------
HAT_TINY = dict(img_size=32, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=16, compress_ratio=3,
                squeeze_factor=6, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=2.0, upscale=4, img_range=1.0, upsampler="pixelshuffle")


def hat_tiny_weights():
    from oracle import hat_oracle as HO
    g = load_golden("g13_hat_tiny")
    cfg = HO.HATConfig(**HAT_TINY)
    sd = HO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g["weight_sha1"]), "HAT weight generator drifted from the one the fixtures were made with"
    return g, cfg, sd


def test_g13_hat_index_tables_bit_exact():
    from oracle import hat_oracle as HO
    g = load_golden("g13_hat_index")
    for ws in (8, 16):
        wse = ws + ws // 2
        assert np.array_equal(HO.rpi_sa(ws), g[f"rpi_sa_ws{ws}"])
        oca = HO.rpi_oca(ws, wse)
        assert np.array_equal(oca, g[f"rpi_oca_ws{ws}"]) and oca.dtype == np.int64
        assert oca.min() < 0                                    # the reference relies on negative-index wrap (hat_arch.py:911-918)
        assert np.array_equal((O.shift_attn_mask(48, 32, ws, ws // 2) != 0).astype(np.uint8), g[f"mask_ws{ws}_48x32"])
    assert int(HO.rpi_oca(16, 24).min()) == -880 and int(HO.rpi_oca(16, 24).max()) == 640


def test_g13_hat_tiny_end_to_end_blocks_and_gradients():
    from oracle import hat_oracle as HO
    g, cfg, sd = hat_tiny_weights()
    for hw in ((32, 32), (32, 48), (20, 37)):
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = HO.hat_forward(sd, cfg, x)
        ref = torch.from_numpy(g[f"y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape == (1, 3, hw[0] * 4, hw[1] * 4)
        assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max())), hw
    xt = torch.from_numpy(g["blk.x"])
    with torch.no_grad():
        p0, p1 = "layers.0.residual_group.blocks.0.", "layers.0.residual_group.blocks.1."
        assert (HO.hab(xt, (32, 48), sd, p1, cfg, 2, 16, 8) - torch.from_numpy(g["blk.hab_shifted"])).abs().max() < 2e-5
        assert (HO.hab(xt, (32, 48), sd, p0, cfg, 2, 16, 0) - torch.from_numpy(g["blk.hab_plain"])).abs().max() < 2e-5
        assert (HO.ocab(xt, (32, 48), sd, "layers.1.residual_group.overlap_attn.", cfg, 2, 16) - torch.from_numpy(g["blk.ocab"])).abs().max() < 2e-5
        assert (HO.forward_features(torch.from_numpy(g["ff.x"]), sd, cfg) - torch.from_numpy(g["ff.y"])).abs().max() < 2e-5
    # autograd over the restatement against the reference's gradients
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(int(g["train.x_seed"])))
    t = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(int(g["train.target_seed"])))
    keys = HO.param_keys(cfg)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    loss = torch.nn.functional.l1_loss(HO.hat_forward({**sd, **leaves}, cfg, x), t)
    grads = dict(zip(keys, torch.autograd.grad(loss, [leaves[k] for k in keys])))
    assert abs(float(loss) - float(g["train.loss"])) < 1e-6
    assert list(g["train.grad_names"]) == keys
    for n, ref in zip(keys, g["train.grad_norms"]):
        assert abs(float(grads[n].norm()) - float(ref)) <= 2e-4 * float(ref) + 1e-9, n
    for k in g.files:
        if k.startswith("grad."):
            ref = torch.from_numpy(g[k])
            assert (grads[k[5:]] - ref).abs().max() <= 1e-5 * max(1e-3, float(ref.abs().max())), k


def test_g13_hat_cfg4_schema_and_probe():
    """HAT-SRx4 (BASELINE cfg4): 20 772 507 parameters / 864 keys; one 64x64 forward against the reference's probes."""
    from oracle import hat_oracle as HO
    g = load_golden("g13_hat_cfg4_probe")
    cfg = HO.HATConfig.sr_x4()
    sd = HO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    assert len(sd) == int(g["n_keys"]) == 864
    assert sum(sd[k].numel() for k in HO.param_keys(cfg)) == int(g["n_params"]) == 20772507
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.no_grad():
        y = HO.hat_forward(sd, cfg, x).numpy()
    assert tuple(y.shape) == tuple(g["shape"])
    assert np.abs(y.reshape(-1)[g["probe_index"]] - g["probe_value"]).max() < 2e-5
    assert abs(float(y.mean()) - float(g["mean"])) < 1e-5 and abs(float(y.std()) - float(g["std"])) < 1e-5


# ---- G14: DAT (reference modules/dat_arch.py; fixtures by oracle/make_golden.py::gen_g14) -----------------------------------
DAT_TINY = dict(img_size=32, in_chans=3, embed_dim=48, split_size=(8, 32), depth=(3, 2), num_heads=(4, 4), expansion_factor=2.0,
                upscale=2, img_range=1.0, resi_connection="1conv", upsampler="pixelshuffle")


def dat_tiny_weights():
    from oracle import dat_oracle as DO
    g = load_golden("g14_dat_tiny")
    cfg = DO.DATConfig(**DAT_TINY)
    sd = DO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g["weight_sha1"]), "DAT weight generator drifted from the one the fixtures were made with"
    return g, cfg, sd


def test_g14_dat_index_tables_bit_exact():
    from oracle import dat_oracle as DO
    g = load_golden("g14_dat_tiny")
    assert np.array_equal(DO.rect_rpi(8, 32), g["rpi_8x32"]) and np.array_equal(DO.rect_rpi(32, 8), g["rpi_32x8"])
    assert np.array_equal(DO.rpe_offsets(8, 32), g["rpe_8x32"])
    assert np.array_equal((DO.rect_shift_mask(32, 64, 8, 32, 4, 16) != 0).astype(np.uint8), g["mask0_32x64"])
    assert np.array_equal((DO.rect_shift_mask(32, 64, 32, 8, 16, 4) != 0).astype(np.uint8), g["mask1_32x64"])
    # img2windows (dat_arch.py:15-23) as a gather map, against the closed form of the reference's view/permute
    idx = DO.rect_window_token_index(16, 64, 8, 32, 0, 0)
    img = np.arange(16 * 64).reshape(16, 64)
    want = img.reshape(2, 8, 2, 32).transpose(0, 2, 1, 3).reshape(4, 256)
    assert np.array_equal(idx, want)
    assert [DO.is_shifted(0, b) for b in range(6)] == [False, False, True, False, False, False]
    assert [DO.is_shifted(1, b) for b in range(6)] == [True, False, False, False, True, False]


def test_g14_dat_tiny_end_to_end_and_blocks():
    from oracle import dat_oracle as DO
    g, cfg, sd = dat_tiny_weights()
    for hw in ((32, 32), (32, 64)):
        x = torch.from_numpy(g[f"x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = DO.dat_forward(sd, cfg, x)
        ref = torch.from_numpy(g[f"y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape == (1, 3, hw[0] * 2, hw[1] * 2)
        assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max())), hw
    xt = torch.randn(1, 32 * 64, 48, generator=torch.Generator().manual_seed(int(g["blk.x_seed"])))
    xs = torch.randn(1, 32 * 32, 48, generator=torch.Generator().manual_seed(int(g["blk.xs_seed"])))
    with torch.no_grad():
        tol = lambda r: 2e-5 * max(1.0, float(torch.from_numpy(r).abs().max()))
        assert (DO.datb(xt, 32, 64, sd, "layers.0.blocks.2.", cfg, 4, 0, 2) - torch.from_numpy(g["blk.spatial_shifted"])).abs().max() < tol(g["blk.spatial_shifted"])
        assert (DO.datb(xs, 32, 32, sd, "layers.0.blocks.0.", cfg, 4, 0, 0) - torch.from_numpy(g["blk.spatial_plain"])).abs().max() < tol(g["blk.spatial_plain"])
        assert (DO.datb(xs, 32, 32, sd, "layers.0.blocks.1.", cfg, 4, 0, 1) - torch.from_numpy(g["blk.channel"])).abs().max() < tol(g["blk.channel"])
        assert (DO.datb(xs, 32, 32, sd, "layers.1.blocks.0.", cfg, 4, 1, 0) - torch.from_numpy(g["blk.shifted_rg1"])).abs().max() < tol(g["blk.shifted_rg1"])
        assert (DO.sgfn(xs, 32, 32, sd, "layers.0.blocks.0.ffn.") - torch.from_numpy(g["blk.sgfn"])).abs().max() < tol(g["blk.sgfn"])


def test_g14_dat_cfg5_schema_and_probe():
    """DAT x4 (BASELINE cfg5): 14 802 051 parameters / 2 116 keys; one 64x64 forward against the reference's probes."""
    from oracle import dat_oracle as DO
    g = load_golden("g14_dat_cfg5_probe")
    cfg = DO.DATConfig.sr_x4()
    sd = DO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    assert len(sd) == int(g["n_keys"]) == 2116
    n_params = sum(v.numel() for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked", "rpe_biases",
                                                                       "relative_position_index", "attn_mask_0", "attn_mask_1")))
    assert n_params == int(g["n_params"]) == 14802051
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    with torch.no_grad():
        y = DO.dat_forward(sd, cfg, x).numpy()
    assert tuple(y.shape) == tuple(g["shape"])
    assert np.abs(y.reshape(-1)[g["probe_index"]] - g["probe_value"]).max() < 2e-5
    assert abs(float(y.mean()) - float(g["mean"])) < 1e-5 and abs(float(y.std()) - float(g["std"])) < 1e-5


# ---- G14b: DAT with padded frames (sizes that are not multiples of the larger split) and split_size [8, 16] ---------------------------
DAT_TINY_816 = dict(DAT_TINY, split_size=(8, 16))


def dat_g14b_weights(tag):
    from oracle import dat_oracle as DO
    g = load_golden("g14b_dat_pad_split")
    cfg = DO.DATConfig(**(DAT_TINY if tag == "s832" else DAT_TINY_816))
    sd = DO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g[f"{tag}.weight_sha1"]), "DAT weight generator drifted from the one the fixtures were made with"
    return g, cfg, sd


@pytest.mark.parametrize("tag", ["s832", "s816"])
def test_g14b_dat_padded_frames_and_128_token_windows(tag):
    """The reference zero-pads q / k / v to a multiple of the larger split (dat_arch.py:376-384): 24x40 and 48x64 inputs with 8x32
    windows, and the reference's own __main__ split [8, 16] (128-token windows) at padded and unpadded sizes; blocks in isolation on
    a padded 24x40 frame incl. the shifted one (masks computed for the padded frame, :404-407)."""
    from oracle import dat_oracle as DO
    g, cfg, sd = dat_g14b_weights(tag)
    sizes = ((24, 40), (48, 64), (32, 32)) if tag == "s832" else ((32, 32), (24, 40), (40, 16))
    for hw in sizes:
        x = torch.from_numpy(g[f"{tag}.x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = DO.dat_forward(sd, cfg, x)
        ref = torch.from_numpy(g[f"{tag}.y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape == (1, 3, hw[0] * 2, hw[1] * 2)
        assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max())), (tag, hw)
    xt = torch.randn(1, 24 * 40, 48, generator=torch.Generator().manual_seed(int(g["blk_seed"])))
    with torch.no_grad():
        for name, b in (("blk_shifted_24x40", 2), ("blk_plain_24x40", 0)):
            ref = torch.from_numpy(g[f"{tag}.{name}"])
            got = DO.datb(xt, 24, 40, sd, f"layers.0.blocks.{b}.", cfg, 4, 0, b)
            assert (got - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max())), (tag, name)


@pytest.mark.parametrize("tag", ["32x32", "24x40"])
def test_g14c_dat_training_step(tag):
    """The training oracle (train_mode: BatchNorm batch statistics + running-statistic updates, autograd for the gradients) against one
    training step of the reference's DAT in .train(): loss, output, all 264 parameter gradients, the 45 BatchNorm buffers."""
    from oracle import dat_oracle as DO
    g = load_golden("g14c_dat_train")
    cfg = DO.DATConfig(**DAT_TINY)
    sd = DO.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g["weight_sha1"])
    x, t = torch.from_numpy(g[f"{tag}.x"]), torch.from_numpy(g[f"{tag}.t"])
    loss, y, grads, rec = DO.loss_and_grads(sd, cfg, x, t)
    assert abs(loss - float(g[f"{tag}.loss"])) <= 1e-6
    assert float((y - torch.from_numpy(g[f"{tag}.y"])).abs().max()) <= 2e-5
    biggest = max(float(torch.from_numpy(g[f"{tag}.grad.{k}"]).norm()) for k in grads)
    for k, v in grads.items():          # gradients that are mathematically zero (a bias in front of a BatchNorm) are 1e-9 noise in both
        r = torch.from_numpy(g[f"{tag}.grad.{k}"])
        assert float((v - r).norm()) <= 1e-3 * float(r.norm()) + 1e-6 * biggest, k
    assert len(rec) == 45
    for k, v in rec.items():
        r = torch.from_numpy(g[f"{tag}.buf.{k}"])
        assert torch.allclose(v.to(r.dtype), r, rtol=1e-5, atol=1e-6), k
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        DO.loss_and_grads(sd, cfg, x[:1], t[:1])


# ---- G16: SwinIR with window_size 16 --------------------------------------------------------------------------------------------------
TINY_W16 = dict(img_size=32, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=16, mlp_ratio=2, img_range=1.0,
                resi_connection="1conv", upscale=2, upsampler="pixelshuffle")


def w16_weights(tag):
    g = load_golden("g16_swinir_w16")
    cfg = O.SwinIRConfig(**dict(TINY_W16, upsampler={"ps": "pixelshuffle", "psd": "pixelshuffledirect"}[tag]))
    sd = O.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g[f"{tag}.weight_sha1"])
    return g, cfg, sd


@pytest.mark.parametrize("tag", ["ps", "psd"])
def test_g16_window_16_forward(tag):
    """The oracle at window_size 16 (256-token windows, 961-row tables, masks recomputed off the training resolution, reflect padding
    to a multiple of 16) against the reference's own SwinIR."""
    g, cfg, sd = w16_weights(tag)
    for hw in ((32, 32), (48, 64), (40, 24)):
        x = torch.from_numpy(g[f"{tag}.x_{hw[0]}x{hw[1]}"])
        with torch.no_grad():
            y = O.swinir_forward(sd, cfg, x)
        ref = torch.from_numpy(g[f"{tag}.y_{hw[0]}x{hw[1]}"])
        assert y.shape == ref.shape and float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), (tag, hw)


# ---- G15: SwinIR with ape=True ------------------------------------------------------------------------------------------------------
TINY_APE = dict(img_size=16, in_chans=3, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=8, mlp_ratio=2, img_range=1.0,
                resi_connection="1conv", upscale=2, upsampler="pixelshuffle", ape=True)


def test_g15_ape_forward_and_gradients():
    g = load_golden("g15_tiny_ape")
    cfg = O.SwinIRConfig(**TINY_APE)
    sd = O.random_state_dict(cfg, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]))
    digest = _sha1(np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()]))
    assert digest == str(g["weight_sha1"])
    with torch.no_grad():
        y = O.swinir_forward(sd, cfg, torch.from_numpy(g["x_16x16"]))
    ref = torch.from_numpy(g["y_16x16"])
    assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
    loss, _, grads = O.loss_and_grads(sd, cfg, torch.from_numpy(g["train.x"]), torch.from_numpy(g["train.target"]))
    assert abs(float(loss) - float(g["train.loss"])) < 1e-5
    for n, gr in grads.items():
        r = torch.from_numpy(g["grad." + n])
        assert (gr - r).abs().max() < 1e-5 * max(1.0, float(r.abs().max())), n
    assert "absolute_pos_embed" in grads
