"""Host-side data path (CPU) and the finetune entry point end-to-end on a synthetic dataset (GPU)."""
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image


def make_dataset(root, n_train=6, n_valid=2, lr=72, scale=4):
    rng = np.random.RandomState(0)
    for split, n in (("train", n_train), ("valid", n_valid)):
        hr_dir = os.path.join(root, "shuffled2D", f"shuffled2D_{split}_HR")
        lr_dir = os.path.join(root, "shuffled2D", f"shuffled2D_{split}_LR_default_X{scale}")
        os.makedirs(hr_dir)
        os.makedirs(lr_dir)
        for i in range(n):
            hr = (rng.rand(lr * scale, lr * scale) * 255).astype(np.uint8)
            Image.fromarray(hr, "L").save(os.path.join(hr_dir, f"{i:04d}.png"))
            Image.fromarray(hr, "L").resize((lr, lr), Image.BICUBIC).save(os.path.join(lr_dir, f"{i:04d}x{scale}.png"))


def test_dataset_pairs_and_transforms(tmp_path):
    from tpu_superresolution_amd.sr_datasets import (PairTransformTrain, PairTransformValid, Shuffled2DPaired, ensure_3ch,
                                                     paired_random_crop)
    make_dataset(str(tmp_path))
    ds = Shuffled2DPaired(str(tmp_path), split="train", scale="X4", transform_pair=PairTransformTrain(64, 4))
    assert len(ds) == 6
    random.seed(1)
    lr, hr = ds[0]
    assert lr.shape == (3, 64, 64) and hr.shape == (3, 256, 256) and lr.dtype == torch.float32
    assert 0.0 <= float(lr.min()) and float(lr.max()) <= 1.0
    vlr, vhr = Shuffled2DPaired(str(tmp_path), split="valid", scale="X4", transform_pair=PairTransformValid(4))[1]
    assert vlr.shape == (3, 72, 72) and vhr.shape == (3, 288, 288)
    # crop alignment: HR crop starts at (top*scale, left*scale)   (finetune_swinir.py:96-110)
    a = torch.arange(10 * 12, dtype=torch.float32).reshape(1, 10, 12)
    b = a.repeat_interleave(2, 1).repeat_interleave(2, 2)
    random.seed(3)
    ca, cb = paired_random_crop(a, b, 4, 2)
    assert torch.equal(cb[:, ::2, ::2], ca)
    with pytest.raises(ValueError):
        paired_random_crop(a, b, 64, 2)
    with pytest.raises(ValueError):
        ensure_3ch(torch.zeros(2, 4, 4))
    with pytest.raises(FileNotFoundError):
        Shuffled2DPaired(str(tmp_path), split="test", scale="X4")


@pytest.mark.gpu
def test_finetune_script_one_epoch(tmp_path, capsys, monkeypatch):
    from tpu_superresolution_amd import finetune_swinir as F
    make_dataset(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    F.main(["--data_root", str(tmp_path), "--scale", "X4", "--epochs", "2", "--batch_size", "2", "--workers", "0", "--lr", "1e-4"])
    out = capsys.readouterr().out
    assert "[X4] epoch 001/2" in out and "[done] best_val_loss=" in out
    ck = torch.load(tmp_path / "best_swinir_finetune_X4.pt", map_location="cpu", weights_only=False)
    assert set(ck) >= {"model", "epoch", "best_val_loss", "val_psnr", "args"}
    assert len(ck["model"]) == 550 and not any(k.startswith("module.") for k in ck["model"])
    # the saved state_dict loads back (strict) through the {"params": ...} envelope of public SwinIR checkpoints
    torch.save({"params": ck["model"]}, tmp_path / "w.pth")
    F.main(["--data_root", str(tmp_path), "--scale", "X4", "--epochs", "1", "--batch_size", "2", "--workers", "0",
            "--weights", str(tmp_path / "w.pth"), "--freeze_regex", "conv_first|layers\\.0"])
    assert "[weights] missing=0, unexpected=0" in capsys.readouterr().out


def test_device_pool_loader_epochs_and_rank_shards():
    """DevicePoolLoader: shuffle + drop_last per epoch, disjoint strided rank shards of one permutation (host logic only)."""
    from tpu_superresolution_amd.finetune_swinir import DevicePoolLoader

    class FakePool:
        def __len__(self):
            return 23

        def sample(self, idx):
            return list(idx)

    seen = []
    for r in range(2):
        ld = DevicePoolLoader(FakePool(), 4, r, 2, seed=7)
        ld.set_epoch(3)
        batches = list(ld)
        assert len(batches) == len(ld) == 2 and all(len(b) == 4 for b in batches)
        seen += sum(batches, [])
    assert len(seen) == len(set(seen)) == 16
    one = DevicePoolLoader(FakePool(), 5)
    e0 = sum(list(one), [])
    one.set_epoch(1)
    e1 = sum(list(one), [])
    assert len(e0) == len(e1) == 20 and e0 != e1 and e0 == sum(list(DevicePoolLoader(FakePool(), 5)), [])


@pytest.mark.gpu
def test_finetune_script_with_the_training_set_on_the_device(tmp_path, capsys, monkeypatch):
    """--gpu_data: pre-decoded pool + device crop instead of the DataLoader; same prints and checkpoint layout."""
    from tpu_superresolution_amd import finetune_swinir as F
    make_dataset(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    F.main(["--data_root", str(tmp_path), "--scale", "X4", "--epochs", "2", "--batch_size", "2", "--workers", "0", "--lr", "1e-4",
            "--gpu_data"])
    out = capsys.readouterr().out
    assert "[gpu_data] 6 pairs" in out and "[X4] epoch 002/2" in out and "[done] best_val_loss=" in out
    ck = torch.load(tmp_path / "bestpsnr_swinir_finetune_X4.pt", map_location="cpu", weights_only=False)
    assert set(ck) >= {"model", "epoch", "best_val_psnr", "val_loss", "args"} and ck["args"]["gpu_data"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["hat", "dat"])
def test_train_and_evaluate_scripts_with_the_transformer_archs(arch, tmp_path, capsys, monkeypatch):
    """train.py / evaluate.py (the reference's entry points, 19 + 10 flags) with the additive --arch hat | dat: one epoch of training on a
    synthetic tree through the HIP training path (DAT: train-mode BatchNorm), the checkpoint it writes evaluated by evaluate.py."""
    from tpu_superresolution_amd import evaluate, train
    root = str(tmp_path / "data")
    make_dataset(root, n_train=4, n_valid=2, lr=48, scale=2)
    hr_dir = os.path.join(root, "shuffled2D", "shuffled2D_test_HR")
    lr_dir = os.path.join(root, "shuffled2D", "shuffled2D_test_LR_default_X2")
    os.makedirs(hr_dir)
    os.makedirs(lr_dir)
    rng = np.random.RandomState(1)
    for i in range(2):
        hr = (rng.rand(96, 96) * 255).astype(np.uint8)
        Image.fromarray(hr, "L").save(os.path.join(hr_dir, f"{i:04d}.png"))
        Image.fromarray(hr, "L").resize((48, 48), Image.BICUBIC).save(os.path.join(lr_dir, f"{i:04d}x2.png"))
    monkeypatch.chdir(tmp_path)
    out = train.main(["--data_root", root, "--scale", "X2", "--epochs", "1", "--batch_size", "2", "--patch_size", "32", "--loss", "l1", "--lr", "1e-4",
                      "--workers", "0", "--arch", arch, "--device", "cuda"])
    text = capsys.readouterr().out
    assert "[X2] epoch 1: train_loss" in text and np.isfinite(out["best"])
    ck = tmp_path / "best_X2.pt"
    assert ck.exists()
    res = evaluate.main(["--scale", "X2", "--data_root", root, "--ckpt", str(ck), "--batch_size", "1", "--save_dir", str(tmp_path / "p"), "--save_n", "1",
                         "--arch", arch, "--device", "cuda"])
    assert np.isfinite(res["psnr"]) and 0 < res["ssim"] <= 1 and res["n"] == 2
