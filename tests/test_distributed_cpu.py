"""world_size-2 gloo rehearsal of the data-parallel gradient path (bucket arithmetic, overlapped
all-reduce bookkeeping, sharding) on CPU tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tpu_superresolution_amd.distributed import GradSynchronizer, merge_buckets, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


def test_merge_buckets_covers_every_segment_once():
    ranges = [(900, 1000), (700, 900), (400, 700), (390, 400), (0, 390)]
    for min_elems in (1, 150, 350, 10 ** 9):
        b = merge_buckets(ranges, min_elems)
        assert b[-1][0] == len(ranges) - 1 and b[0][2] == 1000 and b[-1][1] == 0
        assert all(b[i][1] == b[i + 1][2] for i in range(len(b) - 1))
    with pytest.raises(ValueError):
        merge_buckets([(0, 10), (20, 30)], 1)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ranges = [(900, 1000), (700, 900), (400, 700), (390, 400), (0, 390)]
        flat = torch.zeros(1000)
        sync = GradSynchronizer(ranges, min_bucket_elems=150)
        sync.bind(flat)
        # "backward": each segment writes rank-dependent gradients, then reports completion
        for seg, (b, e) in enumerate(ranges):
            flat[b:e] = torch.arange(b, e, dtype=torch.float32) * (rank + 1)
            sync.segment_done(seg, b, e)
        sync.finish()
        expect = torch.arange(1000, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(flat, expect)
        # data-parallel identity: mean of per-shard mean-gradients == full-batch gradient (equal shards)
        torch.manual_seed(0)
        xs = torch.randn(8, 5)
        w = torch.randn(5, requires_grad=True)
        b, e = shard_range(8, rank, world)
        (xs[b:e] @ w).abs().mean().backward()
        g = w.grad.clone()
        dist.all_reduce(g)
        g /= world
        w2 = w.detach().clone().requires_grad_(True)
        (xs @ w2).abs().mean().backward()
        ok = ok and torch.allclose(g, w2.grad, atol=1e-6)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_synchronizer_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


def _list_worker(rank, world, port, q):
    """ListGradSynchronizer (the HAT backward's per-segment hook): lists of odd-shaped tensors, several segments per step."""
    from tpu_superresolution_amd.distributed import ListGradSynchronizer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sync = ListGradSynchronizer(average=True)
        shapes = [[(3, 4), (7,)], [(2, 2, 3, 3)], [(5,), (1,), (6, 2)]]
        ok = True
        for step in range(2):                  # the synchronizer is reused step after step
            segs = [[torch.full(s, float(rank + 1 + si + step)) for s in seg] for si, seg in enumerate(shapes)]
            for seg in segs:
                sync.segment_done(seg + [None])
            sync.finish()
            for si, seg in enumerate(segs):
                want = sum(r + 1 + si + step for r in range(world)) / world
                ok = ok and all(torch.allclose(t, torch.full_like(t, want)) for t in seg)
            ok = ok and sync.buckets_last_step == len(shapes)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_list_grad_synchronizer_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_list_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}
