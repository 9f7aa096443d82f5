#!/usr/bin/env python
"""Headline benchmark: SwinIR classical x4 TRAIN STEP, 64x64 LR patches, batch 32 per GPU (BASELINE cfg3).

    python bench.py --gpus N --steps K --warmup W                 (N > 1 without WORLD_SIZE: spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config cfg2                                 (SwinIR-light x2 inference, 48x48 LR, bs 16: BASELINE cfg2)
    python bench.py --config cfg4 | cfg5                          (HAT x4 / DAT x4 inference, 64x64 LR, bs 16: BASELINE cfg4 / cfg5)

A step = forward + L1 loss + backward + (N>1: RCCL gradient all-reduce, overlapped) + clip 1.0 + AdamW on
one batch of synthetic LR/HR patches already resident in HBM.  Prints ONE JSON line on rank 0.
Weak scaling: every rank processes its own 32 patches.

The JSON line's `roofline` has the dominant kernel family (HIP events inside the timed region) and `roofline.step`: the
whole step against the MFMA roofline (SURVEY 8d's primary bound) plus HBM bytes per step -- measured (rocprofv3 PMC summary
under profiles/, used only while its kernel-source digest matches the library being run) vs the block-fused compulsory
bytes of SURVEY 8d.

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
FAMILY_TRAFFIC_FILE = "r03_pmc_linear_gemm_stream_traffic.json"     # rocprofv3 PMC summaries this round's library was measured with
STEP_TRAFFIC_FILE = "r03_pmc_step_traffic.json"
HBM_PEAK_GBS = 8000.0                # HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy ceiling)


def synthetic_batch(batch, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lr = torch.rand(batch, 3, 64, 64, generator=g)
    hr = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False)
    hr = (hr + 0.02 * torch.rand(hr.shape, generator=torch.Generator().manual_seed(seed + 1))).clamp(0, 1)
    return lr.to(device), hr.to(device)


def psnr_delta(model, batch=2):
    """BASELINE's quality gate ("PSNR delta vs ref <= 0.01 dB"): the HIP forward and the CPU oracle's fp32 forward of the SAME
    weights (the benchmarked model's state_dict as it stands after the timed steps) on the SURVEY 8d synthetic target."""
    from oracle import swinir_oracle as O
    cfg = O.SwinIRConfig.classical_x4()
    lr, hr = O.synthetic_batch(batch, 64, 4, seed=0)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    was_training = model.training
    model.eval()
    with torch.no_grad():
        y_hip = model(lr.to(next(model.parameters()).device)).float().cpu()
        y_ref = O.swinir_forward(sd, cfg, lr)
    model.train(was_training)
    p_hip, p_ref = float(O.batch_psnr(y_hip, hr).mean()), float(O.batch_psnr(y_ref, hr).mean())
    mse = float(((y_hip - y_ref) ** 2).mean())
    return {"psnr_delta_db": p_hip - p_ref, "psnr_hip_db": p_hip, "psnr_oracle_db": p_ref,
            "max_abs_diff": float((y_hip - y_ref).abs().max()),
            "mutual_psnr_db": float(10.0 * torch.log10(torch.tensor(1.0 / max(mse, 1e-20)))),
            "sample": f"eval forward of the benchmarked weights, batch {batch}, 64x64 LR -> 256x256 HR, HIP bf16 path vs CPU oracle fp32"}


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



def kernels_digest():
    """sha1 over the kernel sources: ties a stored PMC summary to the library it was measured on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "tpu_superresolution_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def stored_traffic(name):
    """-> (value or None, source dict).  A stored rocprofv3 PMC figure is reported only while the kernel sources are the
    ones it was measured on; otherwise the value is withheld (null) and the line says so."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    fresh = d.get("kernels_sha") == kernels_digest()
    src = {"file": "profiles/" + name, "kernels_sha": d.get("kernels_sha"), "matches_this_library": fresh,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, (2*FETCH+WRITE)*1024 (MI355X_MICROARCH.md)"}
    return d, src


def spawn_ranks(n, argv):
    """`bench.py --gpus N` started by hand: become the launcher.  Runs BEFORE anything touches the GPU (no HIP call, no
    torch.cuda.is_available()); the ranks are ordinary child processes (no exec), rank 0's stdout is ours."""
    import socket
    import subprocess
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


# compulsory HBM bytes of a block-fused executor, forward, fp32 residual stream: 301 MB per 64x64 LR image;
# training ~3.25x forward (saved block inputs re-read, dY read, dX written); + 333 MB optimizer traffic per step.
COMPULSORY_FWD_BYTES_PER_IMAGE = 301e6
COMPULSORY_TRAIN_FACTOR = 3.25
OPTIMIZER_BYTES_PER_STEP = 11_900_199 * 28


INFER = {
    "cfg2": dict(metric="HR pixels/sec, SwinIR-light x2 inference, 48x48 LR, bs=16/GPU", batch=16, lr=48, scale=2, flop_per_image=4.818e9,
                 workload="BASELINE cfg2: SwinIR-light x2 (dim 60, 4x6 blocks, window 8, pixelshuffledirect) inference forward, 48x48 LR -> "
                          "96x96 HR, random-init weights",
                 kernel="swin_block_light_kernel (csrc/block_light.hip: one whole Swin block per launch, a window per workgroup) + the 3x3 conv "
                        "implicit GEMMs (gemm_kernel, csrc/gemm.hip)"),
    "cfg4": dict(metric="HR pixels/sec, HAT x4 inference, 64x64 LR, bs=16/GPU", batch=16, lr=64, scale=4, flop_per_image=207.76e9,
                 workload="BASELINE cfg4: HAT-SRx4 (dim 180, 6x6 HAB + 6 OCAB, window 16, overlap 0.5, CAB) inference forward, 64x64 LR -> "
                          "256x256 HR, random-init weights",
                 kernel="linear-layer GEMMs (persistent LDS-DMA GEMMs + fused MLP kernel, csrc/gemm_stream.hip)"),
    "cfg5": dict(metric="HR pixels/sec, DAT x4 inference, 64x64 LR, bs=16/GPU", batch=16, lr=64, scale=4, flop_per_image=136.64e9,
                 workload="BASELINE cfg5: DAT x4 (dim 180, 6x6 blocks alternating 8x32|32x8 spatial and channel attention, SGFN expansion 4) "
                          "inference forward (eval-mode BatchNorm), 64x64 LR -> 256x256 HR, random-init weights",
                 kernel="linear-layer GEMMs (persistent LDS-DMA GEMMs, csrc/gemm_stream.hip)"),
}


def build_infer_model(name, device):
    import tpu_superresolution_amd as T
    torch.manual_seed(42)
    if name == "cfg2":
        m = T.SwinIR(upscale=2, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=[6] * 4, embed_dim=60,
                     num_heads=[6] * 4, mlp_ratio=2, upsampler="pixelshuffledirect")
    elif name == "cfg5":      # official DAT x4 hyper-parameters (the reference only has a DAT-S-like __main__ demo: SURVEY 0)
        m = T.DAT(upscale=4, in_chans=3, img_size=64, img_range=1.0, depth=[6] * 6, embed_dim=180, num_heads=[6] * 6, expansion_factor=4,
                  resi_connection="1conv", split_size=[8, 32], upsampler="pixelshuffle")
    else:       # official HAT-SRx4 hyper-parameters (the reference repo never instantiates HAT: SURVEY 0)
        m = T.HAT(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
                  overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2,
                  upsampler="pixelshuffle", resi_connection="1conv")
    return m.to(device).eval()


def bench_inference(args):
    """BASELINE cfg2 / cfg4: a step = one forward of one batch resident in HBM.  Replicas only for N > 1 (no collective)."""
    import torch.distributed as dist
    from tpu_superresolution_amd import _lib
    from tpu_superresolution_amd.distributed import init_from_env
    spec = INFER[args.config]
    rank, world, local = init_from_env("nccl") if args.gpus > 1 else (0, 1, 0)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    model = build_infer_model(args.config, device)
    bs = args.batch or spec["batch"]
    x = torch.rand(bs, 3, spec["lr"], spec["lr"], generator=torch.Generator().manual_seed(rank)).to(device)
    flop_per_image = spec["flop_per_image"]                    # SURVEY 6 / 8d (FlopCounterMode on the reference)
    hr_px = bs * (spec["lr"] * spec["scale"]) ** 2
    lib = _lib.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):
            y = model(x)
        graph = None
        if not args.no_graph:
            # a few hundred launches per forward: replay them as one hipGraph
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(x)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y = model(x)
            graph.replay()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                y = model(x)
        barrier()
        elapsed = time.perf_counter() - t0
        roof = None
        if rank == 0 and not args.no_roofline:      # un-timed extra pass with every GEMM launch bracketed (eager: events
            _lib.check(lib.srk_set_option(b"probe_stride", 1))       # cannot be recorded inside a captured graph)
            _lib.check(lib.srk_probe_begin(1, 4000))
            for _ in range(5):
                model(x)
            torch.cuda.synchronize()
            ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
            _lib.check(lib.srk_probe_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
            if n.value:
                tfl = fl.value / (ms.value * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": spec["kernel"], "achieved": tfl, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tfl / MFMA_BF16_PEAK_TFLOPS, "traffic": None, "launches": n.value, "avg_launch_us": 1e3 * ms.value / n.value,
                        "share_of_step": (ms.value / 5) / (1e3 * elapsed / args.steps)}
                if by.value > 0:
                    gbs = by.value / (ms.value * 1e-3) / 1e9
                    roof["hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_launch": by.value / n.value}
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    if not bool(torch.isfinite(y).all()):
        raise SystemExit("non-finite output during the benchmark")
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * hr_px * args.steps / elapsed
        tfl_step = bs * flop_per_image / (ms_per_step * 1e-3) / 1e12
        out = {"metric": spec["metric"], "value": value, "unit": "HR pixels/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": spec["workload"], "batch_per_gpu": bs, "global_batch": bs * world, "parallelism": f"replicas{world}",
                          "hip_graph": graph is not None, "per_gpu_value": value / world, "step_tflops_per_gpu": tfl_step}}
        if roof is not None:
            roof["step"] = {"mfma_frac": tfl_step / MFMA_BF16_PEAK_TFLOPS, "algorithmic_tflop_per_step": bs * flop_per_image / 1e12}
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_infer(args.config)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_hat_train(args):
    """BASELINE cfg4 / cfg5 as a TRAIN step: HAT-SRx4 or DAT x4, 64x64 LR, bs 16 per GPU: forward
    (activations kept; DAT: BatchNorm with batch statistics) + L1 + backward (hat_train.hat_backward / dat_train.dat_backward through
    the C ABI) + clip 1.0 + AdamW.  The optimizer is torch's (foreach) AdamW on the module's own parameters -- the fused flat-buffer
    optimizer belongs to the SwinIR engine.  N > 1: per-segment gradient all-reduce on a side stream
    (distributed.ListGradSynchronizer), overlapped with the backward of the next segment; BatchNorm statistics stay per rank (as
    nn.BatchNorm2d under DDP without SyncBatchNorm)."""
    import torch.distributed as dist
    from tpu_superresolution_amd.distributed import ListGradSynchronizer, init_from_env
    from tpu_superresolution_amd.training import l1_loss_checked
    rank, world, local = init_from_env("nccl") if args.gpus > 1 else (0, 1, 0)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    cfg_name = args.config
    model = build_infer_model(cfg_name, device).train()
    if world > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
        model.grad_sync = ListGradSynchronizer()
        model.grad_sync.time_exposed = True
    bs = args.batch or 16
    graphed = world == 1 and not args.no_graph        # the step is launch-bound on the host (hundreds of C-ABI calls + small torch ops): one hipGraph
    opt = torch.optim.AdamW(model.parameters(), lr=2e-5, weight_decay=0.0, capturable=graphed)
    torch.manual_seed(1234 + rank)
    lr_img, hr_img = synthetic_batch(bs, device, seed=1000 + rank)
    if graphed:
        from tpu_superresolution_amd.training import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, max_grad_norm=1.0, warmup=2)

    def step():
        if graphed:
            return gstep(lr_img, hr_img)
        opt.zero_grad(set_to_none=True)
        loss, bad = l1_loss_checked(model(lr_img), hr_img)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss, bad

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        loss, bad = step()
    barrier()
    if world > 1:
        model.grad_sync.exposed_ms()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, bad = step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    if int(bad) or not bool(torch.isfinite(loss)):
        raise SystemExit("non-finite output/loss during the benchmark")
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * bs * HR_PX_PER_SAMPLE * args.steps / elapsed
        flop_step = 3.0 * bs * INFER[cfg_name]["flop_per_image"]            # fwd + bwd ~ 3 x the reference's forward FLOPs
        what = {"cfg4": ("HAT", "BASELINE cfg4 as a train step: HAT-SRx4 (dim 180, 6x6 HAB + 6 OCAB, window 16, overlap 0.5, CAB) fwd + L1 + "),
                "cfg5": ("DAT", "BASELINE cfg5 as a train step: DAT x4 (dim 180, 6x6 DATB, split 8x32, BatchNorm with batch statistics) fwd + L1 + ")}[cfg_name]
        out = {"metric": f"HR pixels/sec, {what[0]} x4 train step, 64x64 LR, bs=16/GPU", "value": value, "unit": "HR pixels/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": what[1] + "bwd + clip 1.0 + AdamW (torch foreach), 64x64 LR -> 256x256 HR, random-init weights, drop_path 0.1",
                          "batch_per_gpu": bs, "global_batch": bs * world, "parallelism": f"dp{world}", "per_gpu_value": value / world,
                          "ms_per_step_median": step_ms[len(step_ms) // 2], "step_tflops_per_gpu": flop_step / (ms_per_step * 1e-3) / 1e12,
                          "launch": "one hipGraph replay per step (training.GraphedTrainStep)" if graphed else "eager launches",
                          "final_loss": float(loss)},
               "roofline": {"bound": "mfma", "kernel": f"whole {what[0]} train step", "achieved": flop_step / (ms_per_step * 1e-3) / 1e12,
                            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop_step / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                            "traffic": None}}
        if world > 1:
            out["dist"] = {"backend": dist.get_backend(), "world_size": world, "buckets": model.grad_sync.buckets_last_step,
                           "allreduce_exposed_ms": model.grad_sync.exposed_ms()}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_infer(name, steps=3):
    spec = INFER[name]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = int(os.environ.get("SRK_CPU_BASELINE_THREADS", max(1, min(avail, 32))))
    torch.set_num_threads(threads)
    if name == "cfg2":
        from oracle import swinir_oracle as O
        cfg, batch = O.SwinIRConfig.light_x2(), 16
        sd = O.random_state_dict(cfg, 42, 1.0)
        fwd = lambda t: O.swinir_forward(sd, cfg, t)
    elif name == "cfg5":
        from oracle import dat_oracle as DO
        cfg, batch = DO.DATConfig.sr_x4(), 2
        sd = DO.random_state_dict(cfg, 42, 1.0)
        fwd = lambda t: DO.dat_forward(sd, cfg, t)
    else:
        from oracle import hat_oracle as HO
        cfg, batch = HO.HATConfig.sr_x4(), 2
        sd = HO.random_state_dict(cfg, 42, 1.0)
        fwd = lambda t: HO.hat_forward(sd, cfg, t)
    x = torch.rand(batch, 3, spec["lr"], spec["lr"], generator=torch.Generator().manual_seed(0))
    times = []
    with torch.no_grad():
        fwd(x)
        for _ in range(steps):
            t0 = time.perf_counter()
            fwd(x)
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": batch * (spec["lr"] * spec["scale"]) ** 2 / med, "unit": "HR pixels/s", "cores": threads, "kind": "port",
            "sample": f"{steps} fp32 forwards of the CPU oracle at batch {batch} (median {med:.2f} s), same model"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["cfg3", "cfg2", "cfg4", "cfg5"], default="cfg3",
                    help="cfg3 (default): the headline train step; cfg2: SwinIR-light x2 / cfg4: HAT x4 / cfg5: DAT x4 inference")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (BASELINE: 32 for cfg3, 16 for cfg2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="cfg2 / --train legs: time the eager launch loop instead of a hipGraph replay")
    ap.add_argument("--train", action="store_true", help="cfg4 / cfg5: time a HAT / DAT TRAIN step (fwd + L1 + bwd + clip + AdamW) instead of inference")
    ap.add_argument("--use-checkpoint", action="store_true",
                    help="cfg3: build the model with use_checkpoint=True (u / h / attention output recomputed in the backward; not the BASELINE config)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="developer A/B: srk_set_option(NAME, VALUE) before the model is built (repeatable)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"cfg3": 50, "cfg2": 200, "cfg4": 10, "cfg5": 10}[args.config]     # >= 50 steps, median reported beside
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])           # never returns
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    for item in args.opt:
        from tpu_superresolution_amd import _lib as _l
        name, _, value = item.partition("=")
        _l.check(_l.lib().srk_set_option(name.encode(), int(value)))
    if args.train:
        if args.config not in ("cfg4", "cfg5"):
            raise SystemExit("--train selects the HAT / DAT train step: use it with --config cfg4 or cfg5 (cfg3, the default, is a train step already)")
        if args.steps == 10:
            args.steps = 5
        return bench_hat_train(args)
    if args.config in INFER:
        return bench_inference(args)
    args.batch = args.batch or 32
    import torch.distributed as dist
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd import _lib
    from tpu_superresolution_amd.distributed import DataParallelSwinIR, init_from_env
    from tpu_superresolution_amd.optim import FusedAdamW
    from tpu_superresolution_amd.training import train_step

    rank, world, local = init_from_env("nccl") if args.gpus > 1 else (0, 1, 0)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    torch.manual_seed(42)                                  # identical random-init weights on every rank
    model = T.SwinIR(upscale=4, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=[6] * 6, embed_dim=180,
                     num_heads=[6] * 6, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv",
                     use_checkpoint=args.use_checkpoint).to(device).train()
    dp = DataParallelSwinIR(model)
    dp.attach(device)
    opt = FusedAdamW(model, lr=2e-5, weight_decay=0.0, max_grad_norm=1.0, grad_div=float(world))
    torch.manual_seed(1234 + rank)                         # DropPath stream differs per rank
    lr_img, hr_img = synthetic_batch(args.batch, device, seed=1000 + rank)
    sync = dp if world > 1 else None
    if world > 1 and dp._sync is not None:
        dp._sync.time_exposed = True

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
    if world > 1 and dp._sync is not None:
        dp._sync.exposed_ms()                              # drop the warm-up steps' samples
    # `ms_per_step` / `value` come from the wall clock around EXACTLY args.steps steps (driver contract); an event per step
    # boundary on the compute stream gives the per-step durations the median is taken from
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, bad = train_step(model, opt, lr_img, hr_img, sync)
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    exposed = dp._sync.exposed_ms() if (world > 1 and dp._sync is not None) else None
    roof = None
    if probe:
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _lib.check(lib.srk_probe_end(C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)))
        if n.value:
            # The linear layers are skinny GEMMs (N, K <= 576): arithmetic intensity 100-150 FLOP/B sits left of the
            # MI355X ridge (~310 FLOP/B), so with one launch per layer the family is HBM-bound, not MFMA-bound.
            gbs = by.value / (ms.value * 1e-3) / 1e9
            tfl = fl.value / (ms.value * 1e-3) / 1e12
            fam, fam_src = stored_traffic(FAMILY_TRAFFIC_FILE)
            traffic = fam["avg_hbm_bytes_per_launch"] if fam and fam_src["matches_this_library"] else None
            roof = {"bound": "hbm", "kernel": "gemm_stream_kernel / gemm_stream_split_kernel / mlp_fused_fwd_kernel (csrc/gemm_stream.hip: "
                                               "persistent LDS-DMA GEMMs of the linear layers -- proj and the fused fc1+GELU+fc2 forward, the "
                                               "fc2 / fc1 / qkv dgrads with their fused bias/GELU'/residual/LayerNorm(-backward) epilogues; the "
                                               "qkv forward lives in attn_fused.hip, the proj dgrad in attn_bwd_fused.hip)",
                    "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": fam_src,
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
               "config": {"workload": ("cfg3 with use_checkpoint=True (NOT the BASELINE line): " if args.use_checkpoint else "") +
                                      "BASELINE cfg3: SwinIR classical x4 (dim 180, 6x6 blocks, window 8) train step = fwd + L1 + bwd "
                                      "+ clip 1.0 + AdamW, 64x64 LR patches -> 256x256 HR, random-init weights, drop_path 0.1",
                          "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                          "per_gpu_value": value / world,
                          "ms_per_step_median": median_ms, "ms_per_step_min": step_ms[0], "ms_per_step_max": step_ms[-1],
                          "timing": "value / ms_per_step: wall clock around the K steps between two barrier + synchronize pairs; "
                                    "median / min / max: HIP events at the step boundaries on the compute stream (rank 0)",
                          "step_tflops_per_gpu": args.batch * FLOP_PER_IMAGE_TRAIN / (ms_per_step * 1e-3) / 1e12,
                          "final_loss": float(loss)}}
        if world > 1:
            s = dp._sync
            out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                           "buckets": len(s.buckets) if s is not None else 0,
                           "bucket_elems": [e - b for _, b, e in s.buckets] if s is not None else [],
                           "allreduce_exposed_ms": exposed,
                           "note": "allreduce_exposed_ms = mean time per step the compute stream waits in finish() after its last "
                                   "backward kernel for the side stream's all-reduce buckets (HIP events, rank 0)"}
        if roof is not None:
            # SURVEY 8d's primary bound is the MFMA roofline of the WHOLE step: that goes first; the dominant kernel family's
            # HBM-roofline view (HIP events around its launches) rides along as `hbm_family`
            comp = args.batch * COMPULSORY_FWD_BYTES_PER_IMAGE * COMPULSORY_TRAIN_FACTOR + OPTIMIZER_BYTES_PER_STEP
            st, st_src = stored_traffic(STEP_TRAFFIC_FILE)
            hbm_step = st["hbm_bytes_per_step"] if st and st_src["matches_this_library"] and args.batch == 32 else None
            tfl_step = out["config"]["step_tflops_per_gpu"]
            out["roofline"] = {"bound": "mfma", "kernel": "whole cfg3 train step (every kernel of fwd + loss + bwd + clip + AdamW; algorithmic, "
                                                          "un-padded FLOPs of SURVEY 8d / BASELINE.md section 2)",
                               "achieved": tfl_step, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl_step / MFMA_BF16_PEAK_TFLOPS,
                               "traffic": hbm_step, "traffic_source": st_src,
                               "step": {"mfma_frac": tfl_step / MFMA_BF16_PEAK_TFLOPS,
                                        "algorithmic_tflop_per_step": args.batch * FLOP_PER_IMAGE_TRAIN / 1e12,
                                        "hbm_bytes_per_step": hbm_step, "hbm_bytes_source": st_src,
                                        "hbm_gbs": (hbm_step / (ms_per_step * 1e-3) / 1e9) if hbm_step else None,
                                        "compulsory_bytes_per_step": comp,
                                        "traffic_ratio": (hbm_step / comp) if hbm_step else None},
                               "hbm_family": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["psnr_delta_db"] = None
            try:
                q = psnr_delta(model)
                out["psnr_delta_db"] = q["psnr_delta_db"]
                out["cpu_baseline"]["quality"] = q
            except Exception as e:      # the quality leg must not take the timing line down
                out["cpu_baseline"]["quality"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
