"""Device data path (srk_paired_crop_u8 through sr_datasets.DevicePairPool) against the host transform it restates
(pil_to_tensor01 + ensure_3ch + paired_random_crop, finetune_swinir.py:80-110): byte/index work, so bit-exact, with the crop
corners drawn from the same `random` state.  SURVEY 8 row f-3, first slice."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from tpu_superresolution_amd import sr_datasets as D


def _pairs(seed, n, scale, gray_every=3, min_lr=20, max_lr=45):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        h, w = int(rng.integers(min_lr, max_lr)), int(rng.integers(min_lr, max_lr))
        if i % gray_every == 0:
            lr = rng.integers(0, 256, (h, w), dtype=np.uint8)
            hr = rng.integers(0, 256, (h * scale, w * scale), dtype=np.uint8)
        else:
            lr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            hr = rng.integers(0, 256, (h * scale, w * scale, 3), dtype=np.uint8)
        out.append((Image.fromarray(lr), Image.fromarray(hr)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("scale,patch", [(4, 16), (2, 20), (3, 7)])
def test_device_batches_equal_the_host_transform(scale, patch):
    pairs = _pairs(scale * 100 + patch, 9, scale)
    pool = D.DevicePairPool(pairs, patch, scale, device="cuda")
    order = [4, 0, 8, 8, 3, 1, 7]                                  # repeats allowed: every draw gets its own corner
    host = D.PairTransformTrain(patch, scale)
    random.seed(1234)
    ref = [host(*pairs[i]) for i in order]
    random.seed(1234)
    lr, hr = pool.sample(order)
    assert lr.shape == (len(order), 3, patch, patch) and hr.shape == (len(order), 3, patch * scale, patch * scale)
    assert torch.equal(lr.cpu(), torch.stack([r[0] for r in ref]))
    assert torch.equal(hr.cpu(), torch.stack([r[1] for r in ref]))
    # both consumed the same number of random draws
    a = random.random()
    random.seed(1234)
    [host(*pairs[i]) for i in order]
    assert a == random.random()


@pytest.mark.gpu
def test_full_image_patch_and_value_range():
    """patch == image size (only corner (0, 0) possible); all 256 byte values map to k / 255 exactly."""
    lr = np.arange(256, dtype=np.uint8).reshape(16, 16)
    hr = np.repeat(np.repeat(lr, 2, 0), 2, 1)
    pool = D.DevicePairPool([(lr, hr)], 16, 2, device="cuda")
    l, h = pool.sample([0])
    want = torch.from_numpy(lr.astype(np.float32) / 255.0)
    assert torch.equal(l.cpu()[0, 0], want) and torch.equal(l.cpu()[0, 2], want)
    assert torch.equal(h.cpu()[0, 1], torch.from_numpy(hr.astype(np.float32) / 255.0))


def test_pool_rejects_what_the_host_transform_rejects():
    small = (np.zeros((8, 8), np.uint8), np.zeros((32, 32), np.uint8))
    with pytest.raises(ValueError, match="too small for patch"):
        D.DevicePairPool([small], 16, 4, device="cpu")
    with pytest.raises(ValueError, match="C=1 or C=3"):
        D.DevicePairPool([(np.zeros((20, 20, 2), np.uint8), np.zeros((80, 80, 2), np.uint8))], 16, 4, device="cpu")
    with pytest.raises(ValueError, match="8-bit and 16-bit"):
        D.DevicePairPool([(np.zeros((20, 20), np.float32), np.zeros((80, 80), np.float32))], 16, 4, device="cpu")
    with pytest.raises(ValueError, match="smaller than scale"):
        D.DevicePairPool([(np.zeros((20, 20), np.uint8), np.zeros((79, 80), np.uint8))], 16, 4, device="cpu")
    pool = D.DevicePairPool([(np.zeros((20, 24), np.uint8), np.zeros((80, 96), np.uint8))], 16, 4, device="cpu")
    assert len(pool) == 1 and pool.pool.numel() == 20 * 24 + 80 * 96


def _pairs16(seed, n, scale):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        h, w = int(rng.integers(20, 40)), int(rng.integers(20, 40))
        if i % 2 == 0:          # 16-bit gray (DeepRock-style "I;16" PNGs) next to 8-bit RGB in one pool
            out.append((rng.integers(0, 65536, (h, w), dtype=np.uint16), rng.integers(0, 65536, (h * scale, w * scale), dtype=np.uint16)))
        else:
            out.append((rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (h * scale, w * scale, 3), dtype=np.uint8)))
    return out


@pytest.mark.gpu
def test_16bit_images_and_sharded_prefetch_equal_the_host_transform():
    """uint16 samples (value / 65535, as pil_to_tensor01) and the pinned-shard pool: batches are bit-identical to the host
    transform and to the single-shard pool, whichever shard is resident; a batch may not straddle shards."""
    scale, patch = 2, 16
    pairs = _pairs16(5, 10, scale)
    host = D.PairTransformTrain(patch, scale)
    whole = D.DevicePairPool(pairs, patch, scale, device="cuda")
    sharded = D.DevicePairPool(pairs, patch, scale, device="cuda", shard_bytes=20000)
    assert whole.num_shards == 1 and sharded.num_shards >= 3
    groups = {}
    for i in range(len(pairs)):
        groups.setdefault(sharded.shard_of(i), []).append(i)
    for s in (2, 0, 1, 2):                                            # out-of-order shard visits: prefetch + switch
        order = groups[s % sharded.num_shards]
        random.seed(99 + s)
        ref = [host(Image.fromarray(pairs[i][0]) if pairs[i][0].dtype == np.uint8 else pairs[i][0],
                    Image.fromarray(pairs[i][1]) if pairs[i][1].dtype == np.uint8 else pairs[i][1]) for i in order]
        random.seed(99 + s)
        l1, h1 = whole.sample(order)
        random.seed(99 + s)
        l2, h2 = sharded.sample(order)
        assert torch.equal(l1, l2) and torch.equal(h1, h2)
        assert torch.equal(l1.cpu(), torch.stack([r[0] for r in ref])) and torch.equal(h1.cpu(), torch.stack([r[1] for r in ref]))
    with pytest.raises(ValueError, match="one shard"):
        sharded.sample([groups[0][0], groups[1][0]])
