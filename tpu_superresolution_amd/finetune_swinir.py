"""SwinIR fine-tuning entry point with the reference's command line (modules/finetune_swinir.py:213-236),
running on MI355X through libsrk.

    python -m tpu_superresolution_amd.finetune_swinir --data_root D --scale X4 --weights swinir.pth [...]
    torchrun --nproc-per-node 8 -m tpu_superresolution_amd.finetune_swinir ...      (data parallel, RCCL)

Same flags, same model configuration (:269-281), same checkpoint envelopes in ({"params": sd} or raw) and
out ("best_swinir_finetune_<scale>.pt", "bestpsnr_swinir_finetune_<scale>.pt" with key "model", :345-371),
same epoch print line (:337-342).  Differences, all additive: bf16 MFMA is built into the kernels (no autocast
context), the step uses the fused L1 / clip / AdamW kernels, `--weights` may be omitted (random init) and
`--drop_path_rate` exposes the constructor default (0.1) that the reference leaves implicit.
"""
from __future__ import annotations

import argparse
import os
import random
import re
import time
from datetime import timedelta

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import SwinIR
from .distributed import DataParallelSwinIR, init_from_env
from .optim import FusedAdamW
from .sr_datasets import PairTransformTrain, PairTransformValid, Shuffled2DPaired
from .training import assert_finite_step, l1_loss, train_step


def fmt(seconds: float) -> str:
    return str(timedelta(seconds=int(seconds)))


def seed_everything(seed: int = 42):
    random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def batch_psnr(pred: torch.Tensor, target: torch.Tensor, max_val: float = 1.0) -> torch.Tensor:
    """finetune_swinir.py:69-74."""
    pred, target = pred.clamp(0.0, 1.0), target.clamp(0.0, 1.0)
    mse = ((pred - target) ** 2).reshape(pred.size(0), -1).mean(dim=1)
    return 20.0 * torch.log10(max_val / torch.sqrt(mse + 1e-8))


def make_loader(ds, batch_size, workers, pin=True, shuffle=False, drop_last=False, persistent=False, sampler=None):
    kw = dict(dataset=ds, batch_size=batch_size, shuffle=shuffle and sampler is None, drop_last=drop_last, num_workers=workers,
              pin_memory=pin, sampler=sampler)
    if workers and workers > 0:
        kw["persistent_workers"] = persistent
        kw["prefetch_factor"] = 2
    return DataLoader(**kw)


class DevicePoolLoader:
    """Drop-in for the training DataLoader when the training set is held pre-decoded in GPU memory (`--gpu_data`,
    sr_datasets.DevicePairPool): same epoch semantics as DataLoader(shuffle=True, drop_last=True) with an optional
    DistributedSampler-style rank shard (per-epoch permutation from torch.Generator(seed + epoch), ranks take strided
    slices of it); the crop corners come from the process-global `random`, as in the host transform."""

    def __init__(self, pool, batch_size: int, rank: int = 0, world: int = 1, seed: int = 0):
        self.pool, self.batch_size, self.rank, self.world, self.seed, self.epoch = pool, batch_size, rank, world, seed, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self):
        return (len(self.pool) // self.world) // self.batch_size

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        perm = torch.randperm(len(self.pool), generator=g).tolist()
        mine = perm[self.rank:len(perm) - len(perm) % self.world:self.world] if self.world > 1 else perm
        if getattr(self.pool, "num_shards", 1) > 1:
            # sharded pool (pinned host shards, two resident on the device): walk the epoch shard by shard -- shuffled order of
            # shards, shuffled images inside each -- so that the copy of the next shard overlaps the steps on this one
            order = torch.randperm(self.pool.num_shards, generator=g).tolist()
            by_shard = {s: [i for i in mine if self.pool.shard_of(i) == s] for s in order}
            for s in order:
                idx = by_shard[s]
                for b in range(len(idx) // self.batch_size):
                    yield self.pool.sample(idx[b * self.batch_size:(b + 1) * self.batch_size])
            return
        for b in range(len(self)):
            yield self.pool.sample(mine[b * self.batch_size:(b + 1) * self.batch_size])


def train_one_epoch(model, loader, optimizer, device, sync=None, check_finite=True):
    model.train()
    total, n, t0 = 0.0, 0, time.time()
    for lr, hr in loader:
        lr, hr = lr.to(device, non_blocking=True), hr.to(device, non_blocking=True)
        loss, bad = train_step(model, optimizer, lr, hr, sync)
        if check_finite:
            assert_finite_step(loss, bad)          # RuntimeError like finetune_swinir.py:133-143
        total += float(loss)
        n += 1
    return total / max(1, n), time.time() - t0


@torch.no_grad()
def validate(model, loader, device):
    model.eval()
    total, n, sum_psnr, n_imgs, t0 = 0.0, 0, 0.0, 0, time.time()
    on_gpu = torch.device(device).type == "cuda"
    if on_gpu:
        # fused L1 + per-image PSNR pass (csrc/misc.hip psnr_*_kernel); the sums stay on the device until the loop ends
        # (the reference reads two scalars back per batch, finetune_swinir.py:196-201)
        from . import ops
        psnr_acc = torch.zeros(1, dtype=torch.float32, device=device)
        l1_acc = torch.zeros(1, dtype=torch.float32, device=device)
    for lr, hr in loader:
        lr, hr = lr.to(device, non_blocking=True), hr.to(device, non_blocking=True)
        out = model(lr)
        if on_gpu:
            batch_abs = torch.zeros(1, dtype=torch.float32, device=device)
            ops.batch_psnr(out.float(), hr.float(), 1.0, psnr_sum=psnr_acc, abs_sum=batch_abs)
            l1_acc += batch_abs / out.numel()          # mean over the batch, like F.l1_loss; batches may differ in size
        else:
            total += float(l1_loss(out, hr))
            sum_psnr += float(batch_psnr(out, hr).sum())
        n += 1
        n_imgs += lr.size(0)
    if on_gpu:
        total, sum_psnr = float(l1_acc), float(psnr_acc)
    return total / max(1, n), sum_psnr / max(1, n_imgs), time.time() - t0


def build_model(scale_int: int, drop_path_rate: float = 0.1) -> SwinIR:
    """finetune_swinir.py:269-281."""
    return SwinIR(upscale=scale_int, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=[6] * 6, embed_dim=180,
                  num_heads=[6] * 6, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv",
                  drop_path_rate=drop_path_rate)


def build_sr_model(arch: str, scale_int: int, drop_path_rate: float = 0.1):
    """The transformer SR models of modules/ at their published x2 / x4 hyper-parameters, on the HIP path: 'swinir'
    (finetune_swinir.py:269-281), 'hat' (HAT-SRx4 configuration of hat_arch.py's constructor defaults: window 16, overlap 0.5, CAB),
    'dat' (official DAT configuration: split [8, 32], expansion 4; dat_arch.py:721-741).  Used by train.py / evaluate.py --arch."""
    if arch == "swinir":
        return build_model(scale_int, drop_path_rate)
    if arch == "hat":
        from .hat_arch import HAT
        return HAT(upscale=scale_int, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
                   overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2,
                   upsampler="pixelshuffle", resi_connection="1conv", drop_path_rate=drop_path_rate)
    if arch == "dat":
        from .dat_arch import DAT
        return DAT(upscale=scale_int, in_chans=3, img_size=64, img_range=1.0, depth=[6] * 6, embed_dim=180, num_heads=[6] * 6,
                   expansion_factor=4, resi_connection="1conv", split_size=[8, 32], upsampler="pixelshuffle", drop_path_rate=drop_path_rate)
    raise ValueError(f"unknown arch {arch!r}")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data_root", type=str, required=True)
    ap.add_argument("--scale", type=str, choices=["X2", "X4"], required=True)
    ap.add_argument("--weights", type=str, default=None, help="Path to SwinIR pretrained checkpoint (.pth/.pt)")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch_size", type=int, default=8, help="per-process batch size")
    ap.add_argument("--lr_patch", type=int, default=64, help="LR patch size (HR patch = lr_patch*scale)")
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no_pin", action="store_true")
    ap.add_argument("--no_persistent", action="store_true")
    ap.add_argument("--freeze_regex", type=str, default=None)
    ap.add_argument("--scheduler", type=str, choices=["None", "Cosine"], default="Cosine")
    ap.add_argument("--min_lr", type=float, default=2e-6)
    ap.add_argument("--grad_clip", type=float, default=1.0)
    ap.add_argument("--drop_path_rate", type=float, default=0.1)      # additive
    ap.add_argument("--gpu_data", action="store_true",
                    help="additive: decode the training set once and crop/convert on the device")
    ap.add_argument("--gpu_data_shard_mb", type=int, default=0,
                    help="additive, with --gpu_data: keep the decoded set in pinned host shards of this size and prefetch them "
                         "to the device one ahead (0 = whole set resident on the device)")
    args = ap.parse_args(argv)

    rank, world, local = init_from_env()
    seed_everything(args.seed)
    if args.workers is None:
        cpu = os.cpu_count() or 4
        args.workers = min(8, max(2, cpu // 2))
    if not torch.cuda.is_available():
        raise SystemExit("the MI355X HIP path needs a GPU (no CPU fallback)")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if rank == 0:
        print("[device]", device, torch.cuda.get_device_name(local), f"world={world}")
    scale_int = 2 if args.scale.upper() == "X2" else 4

    train_ds = Shuffled2DPaired(args.data_root, split="train", scale=args.scale, transform_pair=PairTransformTrain(args.lr_patch, scale_int))
    valid_ds = Shuffled2DPaired(args.data_root, split="valid", scale=args.scale, transform_pair=PairTransformValid(scale_int))
    sampler = DistributedSampler(train_ds, num_replicas=world, rank=rank, shuffle=True, seed=args.seed) if world > 1 else None
    train_loader = make_loader(train_ds, args.batch_size, args.workers, pin=not args.no_pin, shuffle=True, drop_last=True,
                               persistent=not args.no_persistent, sampler=sampler)
    if args.gpu_data:
        from .sr_datasets import DevicePairPool
        raw = Shuffled2DPaired(args.data_root, split="train", scale=args.scale, transform_pair=None)
        pool = DevicePairPool((raw[i] for i in range(len(raw))), args.lr_patch, scale_int, device=device,
                              shard_bytes=(args.gpu_data_shard_mb << 20) or None)
        train_loader = sampler = DevicePoolLoader(pool, args.batch_size, rank, world, args.seed)
        if rank == 0:
            print(f"[gpu_data] {len(pool)} pairs in {pool.num_shards} shard(s), {sum(t.numel() for t in pool._host) / 2**20:.1f} MiB decoded")
    valid_loader = make_loader(valid_ds, max(1, args.batch_size // 2), args.workers, pin=not args.no_pin, shuffle=False,
                               drop_last=False, persistent=not args.no_persistent)

    model = build_model(scale_int, args.drop_path_rate)
    if args.weights:
        ckpt = torch.load(args.weights, map_location="cpu", weights_only=True)
        state = ckpt["params"] if isinstance(ckpt, dict) and "params" in ckpt else (ckpt.get("model", ckpt) if isinstance(ckpt, dict) else ckpt)
        missing, unexpected = model.load_state_dict(state, strict=True)
        if rank == 0:
            print(f"[weights] loaded: {args.weights}")
            print(f"[weights] missing={len(missing)}, unexpected={len(unexpected)}")
    model = model.to(device)
    if args.freeze_regex:
        pattern, froze = re.compile(args.freeze_regex), 0
        for name, p in model.named_parameters():
            if pattern.search(name):
                p.requires_grad = False
                froze += 1
        if rank == 0:
            print(f"[freeze] regex='{args.freeze_regex}', froze_params={froze}")
    if rank == 0:
        n_train = sum(1 for p in model.parameters() if p.requires_grad)
        print(f"[params] trainable tensors: {n_train} / total: {len(list(model.parameters()))}")

    dp = DataParallelSwinIR(model)
    dp.attach(device)                 # weights are now identical on every rank (broadcast from rank 0)
    if world > 1:
        # per-rank randomness from here on: DropPath masks and crop corners must differ between ranks, or stochastic
        # depth / crop diversity would not scale with the world size (bench.py seeds 1234 + rank the same way)
        seed_everything(args.seed + rank)
    opt = FusedAdamW(model, lr=args.lr, weight_decay=args.weight_decay,
                     max_grad_norm=args.grad_clip if args.grad_clip and args.grad_clip > 0 else None, grad_div=float(world))
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.epochs, eta_min=args.min_lr) if args.scheduler == "Cosine" else None

    best_loss, best_psnr, t_all = float("inf"), -float("inf"), time.time()
    for epoch in range(1, args.epochs + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        tr_loss, tr_t = train_one_epoch(model, train_loader, opt, device, dp if world > 1 else None)
        val_loss, val_psnr, val_t = validate(model, valid_loader, device)
        if sched is not None:
            sched.step()
        if rank != 0:
            continue
        print(f"[{args.scale}] epoch {epoch:03d}/{args.epochs} | lr={opt.param_groups[0]['lr']:.2e} | "
              f"train L1={tr_loss:.6f} ({tr_t:.1f}s) | val L1={val_loss:.6f}, PSNR={val_psnr:.2f}dB ({val_t:.1f}s)")
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}     # un-prefixed keys, like the reference
        if val_loss < best_loss:
            best_loss = val_loss
            torch.save({"model": sd, "epoch": epoch, "best_val_loss": best_loss, "val_psnr": val_psnr, "args": vars(args)},
                       f"best_swinir_finetune_{args.scale}.pt")
        if val_psnr > best_psnr:
            best_psnr = val_psnr
            torch.save({"model": sd, "epoch": epoch, "best_val_psnr": best_psnr, "val_loss": val_loss, "args": vars(args)},
                       f"bestpsnr_swinir_finetune_{args.scale}.pt")
    if rank == 0:
        print(f"[time] total: {fmt(time.time() - t_all)}")
        print(f"[done] best_val_loss={best_loss:.6f}, best_val_psnr={best_psnr:.2f} dB")


if __name__ == "__main__":
    main()
