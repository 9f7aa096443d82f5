"""Fused validation metrics (csrc/misc.hip psnr_partial_kernel / psnr_finish_kernel through srk_batch_psnr) against the
oracle's restatement of batch_psnr (finetune_swinir.py:69-74) and its golden vector g12, and the validate() loop that uses
them against the torch-op form it replaces.  SURVEY 8 row f-4, first slice (SSIM stays out: parity unpinned, 8c)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import swinir_oracle as O

pytestmark = pytest.mark.gpu


def test_batch_psnr_matches_golden_g12():
    from tpu_superresolution_amd import ops
    g = load_golden("g12_psnr")
    a, b = torch.from_numpy(g["a"]).float(), torch.from_numpy(g["b"]).float()
    got = ops.batch_psnr(a.cuda(), b.cuda()).cpu()
    assert float((got - torch.from_numpy(g["batch_psnr"])).abs().max()) < 1e-3          # dB


@pytest.mark.parametrize("shape", [(1, 3, 7, 5), (5, 3, 96, 96), (32, 3, 256, 256), (3, 1, 500, 500)])
def test_batch_psnr_and_l1_sum_match_the_oracle(shape):
    """values outside [0, 1] on both sides (the PSNR clamps, the L1 does not), ragged sizes, the cfg3 HR batch, a full
    500 x 500 validation image; repeated calls are bit-identical (fixed summation order)."""
    from tpu_superresolution_amd import ops
    gen = torch.Generator().manual_seed(sum(shape))
    t = torch.rand(shape, generator=gen) * 1.2 - 0.1
    p = t + 0.05 * torch.randn(shape, generator=gen)
    ref = O.batch_psnr(p, t)
    psnr_sum = torch.zeros(1, device="cuda")
    abs_sum = torch.zeros(1, device="cuda")
    got = ops.batch_psnr(p.cuda(), t.cuda(), 1.0, psnr_sum=psnr_sum, abs_sum=abs_sum)
    assert float((got.cpu() - ref).abs().max()) < 2e-3                                   # dB, fp32 sums of up to 250k terms
    assert abs(float(psnr_sum) - float(ref.sum())) < 2e-3 * shape[0]
    l1 = float(abs_sum) / p.numel()
    assert abs(l1 - float(torch.nn.functional.l1_loss(p, t))) < 1e-6
    again = ops.batch_psnr(p.cuda(), t.cuda())
    assert torch.equal(again, got)
    # accumulation semantics: a second batch adds to the running sums
    ops.batch_psnr(p.cuda(), t.cuda(), 1.0, psnr_sum=psnr_sum, abs_sum=abs_sum)
    assert abs(float(psnr_sum) - 2.0 * float(got.sum())) < 1e-3 * shape[0]


def test_identical_images_hit_the_epsilon_floor():
    """mse = 0 -> 20 log10(1 / sqrt(1e-8)) = 80 dB (finetune_swinir.py:73-74)."""
    from tpu_superresolution_amd import ops
    x = torch.rand(2, 3, 16, 16).cuda()
    assert float((ops.batch_psnr(x, x.clone()) - 80.0).abs().max()) < 1e-3


def test_bad_arguments_are_rejected():
    from tpu_superresolution_amd._lib import SrkError, lib
    x = torch.rand(2, 3, 8, 8).cuda()
    ws = torch.empty(64, dtype=torch.uint8, device="cuda")
    out = torch.empty(2, device="cuda")
    rc = lib().srk_batch_psnr(x.data_ptr(), x.data_ptr(), ws.data_ptr(), 0, 192, 1.0, out.data_ptr(), None, None, None)
    assert rc < 0
    rc = lib().srk_batch_psnr(x.data_ptr(), None, ws.data_ptr(), 2, 192, 1.0, out.data_ptr(), None, None, None)
    assert rc < 0
    assert SrkError is not None


def test_validate_loop_matches_the_torch_form():
    """validate() (finetune_swinir.py:181-207) with the fused metrics equals the per-batch torch-op form."""
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd import finetune_swinir as F
    cfg = O.SwinIRConfig(upscale=2, upsampler="pixelshuffle", embed_dim=24, depths=[2], num_heads=[2], window_size=8, mlp_ratio=2,
                         img_size=16)
    sd = O.random_state_dict(cfg, seed=5, scale=1.0)
    m = T.SwinIR(**cfg.kwargs()).cuda().eval()
    m.load_state_dict(sd)
    gen = torch.Generator().manual_seed(0)
    batches = [(torch.rand(3, 3, 16, 16, generator=gen), torch.rand(3, 3, 32, 32, generator=gen)),
               (torch.rand(2, 3, 16, 16, generator=gen), torch.rand(2, 3, 32, 32, generator=gen))]      # ragged last batch
    loss, psnr, _ = F.validate(m, batches, torch.device("cuda"))
    tot, n, ps, ni = 0.0, 0, 0.0, 0
    with torch.no_grad():
        for lr, hr in batches:
            out = m(lr.cuda())
            tot += float(F.l1_loss(out, hr.cuda())); n += 1
            ps += float(F.batch_psnr(out, hr.cuda()).sum()); ni += lr.size(0)
    assert abs(loss - tot / n) < 1e-6
    assert abs(psnr - ps / ni) < 1e-3


@pytest.mark.parametrize("shape", [(3, 1, 50, 37), (2, 3, 128, 96), (1, 1, 11, 11), (4, 1, 500, 500)])
def test_device_ssim_matches_the_torch_operator_form(shape):
    """csrc/metrics.hip srk_ssim against metrics.ssim_torch on the CPU (itself checked against an independent scipy
    evaluation in tests/test_cfg1_plumbing.py).  pytorch_msssim is absent: PARITY UNPINNED against the reference's own SSIM
    numbers -- the algorithm is restated from the published package, not verified against its output."""
    from tpu_superresolution_amd import metrics, ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(*shape, generator=g)
    y = (x + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
    want = metrics.ssim_torch(x, y, data_range=1.0, size_average=False)
    per, mean = ops.ssim(x.cuda(), y.cuda(), 1.0)
    assert float((per.cpu() - want).abs().max()) <= 2e-5
    assert abs(float(mean) - float(want.mean())) <= 2e-5
    per2, _ = ops.ssim(x.cuda(), y.cuda(), 1.0)
    assert torch.equal(per, per2)                                        # fixed-order sums
    # the module-level entry dispatches GPU tensors to the kernel, both reductions
    assert abs(float(metrics.ssim(x.cuda(), y.cuda(), data_range=1.0)) - float(want.mean())) <= 2e-5
    assert float((metrics.ssim(x.cuda(), y.cuda(), data_range=1.0, size_average=False).cpu() - want).abs().max()) <= 2e-5
    assert float(metrics.ssim(x.cuda(), x.cuda(), data_range=1.0)) == pytest.approx(1.0, abs=1e-6)
    # data_range 255 (the published default) scales C1 / C2
    w255 = metrics.ssim_torch(x * 255, y * 255, data_range=255.0, size_average=True)
    assert abs(float(metrics.ssim((x * 255).cuda(), (y * 255).cuda())) - float(w255)) <= 2e-5


def test_device_eval_psnr_matches_evaluate_formula():
    """evaluate.py:24-29 (no clamp, mse floored at 1e-10, batch mean) on the device; values from golden G12 ("restated from
    text" by oracle/make_golden.py, not reference output: torchvision / pytorch_msssim keep evaluate.py from importing)."""
    from tpu_superresolution_amd import metrics, ops
    g = load_golden("g12_psnr")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    per, mean = ops.eval_psnr(a.cuda(), b.cuda(), 1.0)
    assert abs(float(mean) - float(g["eval_psnr"])) <= 1e-4
    assert abs(float(mean) - O.eval_psnr(a, b)) <= 1e-4
    assert abs(metrics.psnr(a.cuda(), b.cuda()) - float(g["eval_psnr"])) <= 1e-4
    assert float(ops.eval_psnr(a.cuda(), a.cuda())[1]) == pytest.approx(100.0, abs=1e-3)      # floor 1e-10 -> 100 dB
    assert float((metrics.batch_psnr(a.cuda(), b.cuda()).cpu() - torch.from_numpy(g["batch_psnr"])).abs().max()) <= 1e-4
