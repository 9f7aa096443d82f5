"""One rank of the data-parallel rehearsal (tests/test_gpu_distributed.py): the REAL engine path -- DataParallelSwinIR.attach
(weight broadcast), the C backward's per-segment hook, GradSynchronizer's GPU branch (event, side stream, async all_reduce)
and FusedAdamW(grad_div=world) -- with world ranks sharing cuda:0 over gloo (a one-GPU box cannot run RCCL between two
ranks: it refuses duplicate devices).  Rank r trains on its shard of a fixed batch; rank 0 writes the reduced flat gradient
and the post-step weights for the parent to compare with a single-process run over the whole batch.

    RANK/WORLD_SIZE/MASTER_* from the env;  argv: <out.pt>
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rehearsal_case():
    from oracle import swinir_oracle as O
    cfg = O.SwinIRConfig(upscale=2, in_chans=3, img_size=64, window_size=8, img_range=1.0, depths=(2, 2), embed_dim=180,
                         num_heads=(6, 6), mlp_ratio=2.0, upsampler="pixelshuffle", resi_connection="1conv")
    sd = O.random_state_dict(cfg, seed=7, scale=1.0)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(8, 3, 64, 64, generator=gen)
    t = torch.rand(8, 3, 128, 128, generator=gen)
    return cfg, sd, x, t


def run(rank, world, device, out_path=None):
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd.distributed import DataParallelSwinIR, shard_range
    from tpu_superresolution_amd.optim import FusedAdamW
    from tpu_superresolution_amd.training import train_step
    cfg, sd, x, t = rehearsal_case()
    model = T.SwinIR(drop_path_rate=0.0, **cfg.kwargs())
    if rank == 0:
        model.load_state_dict(sd, strict=True)          # other ranks keep their own random init: attach() must broadcast
    model = model.to(device).train()
    dp = DataParallelSwinIR(model, min_bucket_elems=1 << 18)
    dp.attach(device)
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0, grad_div=float(world))
    b, e = shard_range(x.shape[0], rank, world)
    loss, bad = train_step(model, opt, x[b:e].to(device), t[b:e].to(device), dp if world > 1 else None)
    torch.cuda.synchronize()
    if rank == 0 and out_path:
        eng = model._engine
        torch.save({"grad": eng.flat_grad.cpu(), "flat": eng.flat.cpu(), "loss": float(loss), "bad": int(bad),
                    "buckets": len(dp._sync.buckets) if dp._sync else 0}, out_path)


def main():
    from tpu_superresolution_amd.distributed import init_from_env
    rank, world, local = init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    run(rank, world, device, sys.argv[1] if len(sys.argv) > 1 else None)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
