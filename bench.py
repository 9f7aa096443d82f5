#!/usr/bin/env python
"""Headline benchmark: SwinIR classical x4 TRAIN STEP, 64x64 LR patches, batch 32 per GPU (BASELINE cfg3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + L1 loss + backward + (N>1: RCCL gradient all-reduce, overlapped) + clip 1.0 + AdamW on
one batch of synthetic LR/HR patches already resident in HBM.  Prints ONE JSON line on rank 0.
Weak scaling: every rank processes its own 32 patches.

Extra legs (rank 0, N=1):
  roofline     -- HIP-event timing of the dominant kernel family (the linear-layer MFMA GEMM) bracketed
                  inside the timed steps by libsrk's probe; achieved = algorithmic bytes / kernel time (HBM-bound
                  family), with the MFMA view alongside.
  cpu_baseline -- the CPU oracle (oracle/swinir_oracle.py, a port of the reference) running the same
                  train step at batch 2 on the host cores.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HR_PX_PER_SAMPLE = 256 * 256
FLOP_PER_IMAGE_TRAIN = 321.299e9      # fwd+bwd algorithmic FLOPs per 64x64 LR image
MFMA_BF16_PEAK_TFLOPS = 2500.0       # MI355X dense bf16 (MI355X_MICROARCH.md)
PROBE_STRIDE = 5
HBM_PEAK_GBS = 8000.0                # HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy ceiling)


def synthetic_batch(batch, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lr = torch.rand(batch, 3, 64, 64, generator=g)
    hr = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False)
    hr = (hr + 0.02 * torch.rand(hr.shape, generator=torch.Generator().manual_seed(seed + 1))).clamp(0, 1)
    return lr.to(device), hr.to(device)


def cpu_baseline(steps=3, batch=2, threads=None):
    """Reference algorithm (CPU oracle, fp32) on the host cores: same train step, bounded sample."""
    from oracle import swinir_oracle as O
    if threads is None:
        try:
            avail = len(os.sched_getaffinity(0))          # the box's CPU share, not the host's core count
        except AttributeError:
            avail = os.cpu_count() or 1
        threads = int(os.environ.get("SRK_CPU_BASELINE_THREADS", max(1, min(avail, 32))))
    torch.set_num_threads(threads)
    cfg = O.SwinIRConfig.classical_x4()
    state = O.TrainState(sd=O.random_state_dict(cfg, 42, 1.5))
    lr, hr = O.synthetic_batch(batch, 64, 4, seed=0)
    O.train_step(state, cfg, lr, hr, lr=2e-5, wd=0.0, grad_clip=1.0)      # warm-up
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        O.train_step(state, cfg, lr, hr, lr=2e-5, wd=0.0, grad_clip=1.0)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": batch * HR_PX_PER_SAMPLE / med, "unit": "HR pixels/s", "cores": threads, "kind": "port",
            "sample": f"{steps} fp32 train steps of the CPU oracle at batch {batch} (median {med:.2f} s/step), same model/loss/optimizer"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU (BASELINE: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd import _lib
    from tpu_superresolution_amd.distributed import DataParallelSwinIR, init_from_env
    from tpu_superresolution_amd.optim import FusedAdamW
    from tpu_superresolution_amd.training import train_step

    rank, world, local = init_from_env("nccl") if args.gpus > 1 else (0, 1, 0)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    torch.manual_seed(42)                                  # identical random-init weights on every rank
    model = T.SwinIR(upscale=4, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=[6] * 6, embed_dim=180,
                     num_heads=[6] * 6, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv").to(device).train()
    dp = DataParallelSwinIR(model)
    dp.attach(device)
    opt = FusedAdamW(model, lr=2e-5, weight_decay=0.0, max_grad_norm=1.0, grad_div=float(world))
    torch.manual_seed(1234 + rank)                         # DropPath stream differs per rank
    lr_img, hr_img = synthetic_batch(args.batch, device, seed=1000 + rank)
    sync = dp if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, bad = train_step(model, opt, lr_img, hr_img, sync)
    barrier()

    probe = (not args.no_roofline) and rank == 0
    lib = _lib.lib()
    if probe:
        # family 1 = linear-layer GEMMs (7 launches per Swin block; 8 where the fused qkv+attention kernel does not apply).
        # Every 5th launch is bracketed by HIP events on the launch stream: 5 is coprime with both block patterns, so the
        # sample is uniform over the kernels; bracketing every launch costs ~4 % of the step in event records.
        _lib.check(lib.srk_set_option(b"probe_stride", PROBE_STRIDE))
        _lib.check(lib.srk_probe_begin(1, 400 * args.steps))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, bad = train_step(model, opt, lr_img, hr_img, sync)
    barrier()
    elapsed = time.perf_counter() - t0
    roof = None
    if probe:
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _lib.check(lib.srk_probe_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
        if n.value:
            # The linear layers are skinny GEMMs (N, K <= 576): arithmetic intensity 100-150 FLOP/B sits left of the
            # MI355X ridge (~310 FLOP/B), so with one launch per layer the family is HBM-bound, not MFMA-bound.
            gbs = by.value / (ms.value * 1e-3) / 1e9
            tfl = fl.value / (ms.value * 1e-3) / 1e12
            traffic = None
            tf = os.path.join(ROOT, "profiles", "r01_pmc_linear_gemm_stream_traffic.json")
            if os.path.exists(tf):
                traffic = json.load(open(tf))["avg_hbm_bytes_per_launch"]      # rocprofv3 PMC, same workload
            roof = {"bound": "hbm", "kernel": "gemm_stream_kernel / gemm_stream_split_kernel (csrc/gemm_stream.hip: persistent LDS-DMA "
                                               "GEMM of the linear layers -- proj/fc1/fc2 forward and the qkv/proj/fc1/fc2 dgrads with their "
                                               "fused bias/GELU/residual/LayerNorm epilogues; the qkv forward lives in attn_fused.hip)",
                    "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": by.value / n.value, "launches": n.value, "sampled_every": PROBE_STRIDE,
                    "avg_launch_us": 1e3 * ms.value / n.value, "share_of_step": PROBE_STRIDE * ms.value / (1e3 * elapsed),
                    "mfma": {"achieved": tfl, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_BF16_PEAK_TFLOPS}}

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    if int(bad) or not bool(torch.isfinite(loss)):
        raise SystemExit("non-finite output/loss during the benchmark")

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.batch * HR_PX_PER_SAMPLE * args.steps / elapsed
        out = {"metric": "HR pixels/sec, SwinIR x4 train step, 64x64 LR, bs=32/GPU", "value": value, "unit": "HR pixels/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "BASELINE cfg3: SwinIR classical x4 (dim 180, 6x6 blocks, window 8) train step = fwd + L1 + bwd "
                                      "+ clip 1.0 + AdamW, 64x64 LR patches -> 256x256 HR, random-init weights, drop_path 0.1",
                          "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                          "per_gpu_value": value / world,
                          "step_tflops_per_gpu": args.batch * FLOP_PER_IMAGE_TRAIN / (ms_per_step * 1e-3) / 1e12,
                          "final_loss": float(loss)}}
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
