"""BASELINE config 1 (CPU plumbing): MS_ResUNet vs golden G11 (outputs of the imported reference), the PIL-only pair
transforms, the metrics, and the evaluate.py / train.py entry points end to end on a synthetic shuffled2D tree.

Pinned by reference output: the MS_ResUNet graph + state_dict schema (G11).  Restated from text, parity unpinned
(torchvision / pytorch_msssim are not importable here): the transforms and SSIM -- they are checked against
independent closed forms (PIL identities, scipy filtering), not against reference-produced fixtures."""
import hashlib
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden
from oracle.cfg1_weights import fill_state_dict

import tpu_superresolution_amd as T
from tpu_superresolution_amd import metrics, sr_transforms as TR


def _sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ms_model():
    g = load_golden("g11_ms_resunet")
    torch.manual_seed(0)
    m = T.MS_ResUNet()
    sd = fill_state_dict(m.state_dict(), seed=int(g["weight_seed"]))
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return g, m, sd


def test_ms_resunet_schema_and_param_count(ms_model):
    g, m, _ = ms_model
    sd = m.state_dict()
    assert len(sd) == int(g["n_keys"]) == 360                                        # model_debug.ipynb cell 6-7
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 24918369
    digest = hashlib.sha1("\n".join(f"{k}:{tuple(v.shape)}:{v.dtype}" for k, v in sd.items()).encode()).hexdigest()
    assert digest == str(g["schema_sha1"])          # same keys, shapes, dtypes, in the reference's order
    assert T.MSResUNet is T.MS_ResUNet


@pytest.mark.parametrize("tag", ["128", "odd"])
def test_ms_resunet_eval_forward_vs_reference_golden(ms_model, tag):
    g, m, _ = ms_model
    m.eval()
    shape = tuple(int(v) for v in g[f"{tag}.shape"])
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(int(g[f"{tag}.input_seed"])))
    with torch.no_grad():
        y = m(x).numpy()
    assert y.shape == shape                       # same-size in/out, incl. the odd size that needs _crop_like
    idx = g[f"{tag}.probe_index"]
    assert np.abs(y.reshape(-1)[idx] - g[f"{tag}.probe_value"]).max() <= 1e-5
    assert abs(float(y.mean()) - float(g[f"{tag}.mean"])) <= 1e-6 and abs(float(y.std()) - float(g[f"{tag}.std"])) <= 1e-6


def test_ms_resunet_train_mode_loss_and_gradients_vs_reference_golden(ms_model):
    g, _, sd = ms_model
    m = T.MS_ResUNet()
    m.load_state_dict(sd, strict=True)
    m.train()
    x = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(3))
    t = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(4))
    loss = torch.nn.functional.mse_loss(m(x), t)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["train.loss"])) <= 1e-5 * abs(float(g["train.loss"]))
    named = dict(m.named_parameters())
    for n, ref in zip(g["train.grad_names"], g["train.grad_norms"]):
        got = float(named[str(n)].grad.norm())
        assert abs(got - float(ref)) <= 1e-4 * float(ref) + 1e-9, n
    assert np.abs(m.bn1.running_mean.numpy() - g["train.running_mean_bn1"]).max() <= 1e-6   # BatchNorm momentum update


# ---- transforms -------------------------------------------------------------------------------------------------
def _rgb(h, w, seed):
    return Image.fromarray(np.random.RandomState(seed).randint(0, 256, size=(h, w, 3), dtype=np.uint8))


def test_eval_transform_is_gray_bicubic_upscale_to_tensor():
    lr, hr = _rgb(16, 12, 0), _rgb(32, 24, 1)
    a, b = TR.build_pair_transform_eval()(lr, hr)
    assert a.shape == b.shape == (1, 32, 24) and a.dtype == b.dtype == torch.float32
    assert torch.equal(b, torch.from_numpy(np.asarray(hr.convert("L"), dtype=np.float32) / 255.0)[None])
    want = np.asarray(lr.convert("L").resize((24, 32), Image.BICUBIC), dtype=np.float32) / 255.0
    assert torch.equal(a, torch.from_numpy(want)[None])
    assert 0.0 <= float(a.min()) and float(a.max()) <= 1.0


def test_gray_passthrough_and_tensor_path():
    g = TR.PairGrayscale()
    l = Image.fromarray(np.zeros((4, 4), np.uint8), mode="L")
    assert g.gray(l) is l
    i16 = Image.fromarray(np.full((4, 4), 40000, np.uint16))
    assert g.gray(i16) is i16
    t = torch.rand(3, 5, 5)
    want = 0.2989 * t[0] + 0.587 * t[1] + 0.114 * t[2]
    assert torch.allclose(g.gray(t)[0], want) and g.gray(t).shape == (1, 5, 5)
    assert g.gray(torch.rand(5, 5)).shape == (1, 5, 5)
    assert TR.to_tensor01(i16).max() == pytest.approx(40000 / 65535.0)


def test_random_crop_is_paired_seeded_and_handles_small_images():
    hr = _rgb(40, 50, 2).convert("L")
    lr = hr.copy()
    torch.manual_seed(7)
    a, b = TR.PairRandomCrop(16)(lr, hr)
    assert a.size == b.size == (16, 16) and np.array_equal(np.asarray(a), np.asarray(b))
    torch.manual_seed(7)
    top, left = int(torch.randint(0, 40 - 16 + 1, (1,))), int(torch.randint(0, 50 - 16 + 1, (1,)))     # top first, then left
    assert np.array_equal(np.asarray(a), np.asarray(hr)[top:top + 16, left:left + 16])
    same = TR.PairRandomCrop((40, 50))(lr, hr)
    assert same[0] is lr and same[1] is hr
    small = TR.PairRandomCrop(64)(lr, hr)                       # patch larger than the image: centre crop to min size
    assert small[0].size == (50, 40)
    ta, tb = TR.PairRandomCrop(8)(torch.arange(400.).view(1, 20, 20), torch.arange(400.).view(1, 20, 20))
    assert ta.shape == (1, 8, 8) and torch.equal(ta, tb)


def test_flips_follow_the_two_uniform_draws():
    img = _rgb(6, 7, 3).convert("L")
    for seed in range(6):
        torch.manual_seed(seed)
        h, v = bool(torch.rand(()) < 0.5), bool(torch.rand(()) < 0.5)
        torch.manual_seed(seed)
        a, b = TR.PairFlips()(img, img)
        want = np.asarray(img)
        want = want[:, ::-1] if h else want
        want = want[::-1] if v else want
        assert np.array_equal(np.asarray(a), want) and np.array_equal(np.asarray(b), want)


def test_train_transform_pipeline_shapes():
    tf = TR.build_pair_transform(patch_size=24, do_flips=True)
    a, b = tf(_rgb(20, 20, 4), _rgb(40, 40, 5))
    assert a.shape == b.shape == (1, 24, 24)


# ---- metrics ------------------------------------------------------------------------------------------------------
def test_psnr_formulas_match_golden_g12():
    g = load_golden("g12_psnr")      # restated from text by oracle/make_golden.py (not reference output)
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    assert np.abs(metrics.batch_psnr(a, b).numpy() - g["batch_psnr"]).max() <= 1e-4
    assert abs(metrics.psnr(a, b) - float(g["eval_psnr"])) <= 1e-4
    assert metrics.psnr(a, a) == pytest.approx(100.0, abs=1e-3)                # mse floor 1e-10 -> 100 dB


def test_ssim_against_an_independent_scipy_form():
    """parity unpinned (pytorch_msssim absent): compared with the published definition evaluated with scipy filtering."""
    from scipy.ndimage import correlate1d
    rs = np.random.RandomState(0)
    X = rs.rand(2, 2, 30, 26).astype(np.float64)
    Y = np.clip(X + 0.1 * rs.randn(*X.shape), 0, 1)
    coords = np.arange(11) - 5
    w = np.exp(-coords ** 2 / (2 * 1.5 ** 2))
    w /= w.sum()                 # fp64 here; the published window is built in fp32 and then cast -> agreement to ~1e-7

    def blur(a):
        a = correlate1d(a, w, axis=2, mode="constant")[:, :, 5:-5, :]
        return correlate1d(a, w, axis=3, mode="constant")[:, :, :, 5:-5]
    mu1, mu2 = blur(X), blur(Y)
    s11, s22, s12 = blur(X * X) - mu1 ** 2, blur(Y * Y) - mu2 ** 2, blur(X * Y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * ((2 * s12 + C2) / (s11 + s22 + C2))
    want_per_image = smap.reshape(2, 2, -1).mean(-1).mean(1)
    got = metrics.ssim(torch.from_numpy(X), torch.from_numpy(Y), data_range=1.0, size_average=False).numpy()
    assert np.abs(got - want_per_image).max() <= 1e-6
    got32 = float(metrics.ssim(torch.from_numpy(X).float(), torch.from_numpy(Y).float(), data_range=1.0))
    assert abs(got32 - want_per_image.mean()) <= 1e-5
    assert float(metrics.ssim(torch.from_numpy(X), torch.from_numpy(X), data_range=1.0)) == pytest.approx(1.0, abs=1e-12)
    with pytest.raises(ValueError):
        metrics.ssim(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 9))


# ---- entry points -----------------------------------------------------------------------------------------------------
def _make_tree(root, split, n, hr_size, scale_tag="X2", seed=0):
    hr_dir = os.path.join(root, "shuffled2D", f"shuffled2D_{split}_HR")
    lr_dir = os.path.join(root, "shuffled2D", f"shuffled2D_{split}_LR_default_{scale_tag}")
    os.makedirs(hr_dir), os.makedirs(lr_dir)
    s = int(scale_tag[1:])
    rs = np.random.RandomState(seed)
    for i in range(n):
        base = rs.rand(hr_size // 4 + 1, hr_size // 4 + 1, 3)
        hr = np.asarray(Image.fromarray((base * 255).astype(np.uint8)).resize((hr_size, hr_size), Image.BICUBIC))
        Image.fromarray(hr).save(os.path.join(hr_dir, f"{i:04d}.png"))
        Image.fromarray(hr).resize((hr_size // s, hr_size // s), Image.BICUBIC).save(os.path.join(lr_dir, f"{i:04d}x{s}.png"))


def test_evaluate_main_on_a_synthetic_test_tree(tmp_path, ms_model, capsys):
    from tpu_superresolution_amd import evaluate
    _, _, sd = ms_model
    root = str(tmp_path / "data")
    _make_tree(root, "test", 5, 32)
    ck = str(tmp_path / "ck.pt")
    torch.save({"model": sd}, ck)
    out = evaluate.main(["--scale", "X2", "--data_root", root, "--ckpt", ck, "--batch_size", "2", "--save_dir", str(tmp_path / "p"),
                         "--save_indices", "0,3,4", "--save_n", "2", "--device", "cpu"])
    text = capsys.readouterr().out
    assert "[data] test samples: 5 | steps: 3" in text and "[ckpt] loaded state_dict from 'model' key" in text
    assert "[baseline] Bicubic PSNR:" in text and "[done] test PSNR:" in text and "[save] explicit indices: [0, 3, 4]" in text
    files = sorted(os.listdir(tmp_path / "p"))
    assert files == [f"idx_{i:06d}_{k}.png" for i in (0, 3) for k in ("hr", "lr", "sr")]        # capped by --save_n 2
    assert Image.open(tmp_path / "p" / "idx_000000_sr.png").size == (32, 32)
    assert np.isfinite(out["psnr"]) and 0 < out["ssim"] <= 1 and out["n"] == 5
    # the bicubic baseline is the eval transform's upscaled LR against HR: recompute it independently
    ds = evaluate.Shuffled2DPaired(root, split="test", scale="X2", transform_pair=TR.build_pair_transform_eval())
    batches = [[ds[i] for i in idx] for idx in ((0, 1), (2, 3), (4,))]
    want = np.mean([metrics.psnr(torch.stack([p[0] for p in b]), torch.stack([p[1] for p in b])) for b in batches])
    assert abs(out["bicubic_psnr"] - want) <= 1e-5
    # raw state_dict envelope + the periodic policy
    torch.save(sd, ck)
    evaluate.main(["--scale", "X2", "--data_root", root, "--ckpt", ck, "--batch_size", "4", "--save_dir", str(tmp_path / "q"),
                   "--save_every", "2", "--save_start", "1", "--device", "cpu"])
    assert "[ckpt] loaded raw state_dict" in capsys.readouterr().out
    assert sorted({f[:10] for f in os.listdir(tmp_path / "q")}) == ["idx_000001", "idx_000003"]


def test_train_main_one_epoch_then_resume(tmp_path, monkeypatch, capsys):
    from tpu_superresolution_amd import train
    root = str(tmp_path / "data")
    _make_tree(root, "train", 4, 24, seed=1)
    _make_tree(root, "valid", 2, 24, seed=2)
    monkeypatch.chdir(tmp_path)
    common = ["--data_root", root, "--scale", "X2", "--batch_size", "2", "--patch_size", "16", "--workers", "0", "--device", "cpu",
              "--loss", "l1"]
    out = train.main(common + ["--epochs", "1"])
    text = capsys.readouterr().out
    assert "[profile X2 loader]" in text and "[X2] epoch 1: train_loss" in text and "val_SSIM" in text
    assert "[plot] saved loss curves to loss_curve_X2.png" in text and os.path.exists(tmp_path / "loss_curve_X2.png")
    ck = torch.load(tmp_path / "best_X2.pt", map_location="cpu", weights_only=True)
    assert set(ck) == {"model", "opt", "sched", "epoch", "args"} and ck["sched"] is None and ck["epoch"] == 1
    assert len(ck["model"]) == 360 and np.isfinite(out["best"])
    # resume restores the optimizer; fine-tune with a freeze regex does not
    train.main(common + ["--epochs", "1", "--resume", str(tmp_path / "best_X2.pt"), "--scheduler", "Exponential"])
    text = capsys.readouterr().out
    assert "[ckpt] loaded model weights from" in text and "[ckpt] restored optimizer state" in text and "[sched] ExponentialLR" in text
    train.main(common + ["--epochs", "1", "--resume", str(tmp_path / "best_X2.pt"), "--finetune", "--freeze_regex", "layer|bn1",
                         "--ft_lr", "1e-5", "--scheduler", "OneCycle"])
    text = capsys.readouterr().out
    assert "[finetune] froze params matching regex: layer|bn1" in text and "restored optimizer state" not in text
