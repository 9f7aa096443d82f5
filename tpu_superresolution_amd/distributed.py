"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code; this is the MI355X-side design for BASELINE cfg3 (LR/HR
patch batches sharded across ranks, gradients summed).  SwinIR has no cross-sample
operator (LayerNorm only), so pure data parallelism is exact.

The C backward runs in segments (tail, RSTB L-1 .. RSTB 0, head); the parameters -- and therefore
the gradients -- of a segment are one contiguous slice of the flat fp32 gradient buffer.  As soon as
a segment has been enqueued on the compute stream, its slice is all-reduced on a side HIP stream
(event-ordered after the segment), so the collective of segment s overlaps the backward kernels of
segment s+1.  ``finish()`` joins the side stream before the optimizer reads the gradients (the global
norm must see the reduced gradients).  Gradients are summed; the division by world size is folded
into the fused optimizer (``FusedAdamW.grad_div``).

``GradSynchronizer`` is device-agnostic (CPU tensors + gloo work too) so that the bucket arithmetic
and the collective pattern are covered by world_size-2 CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  backend 'nccl' is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SRK_SHARE_GPU") == "1":
        local = 0            # rehearsal on a one-GPU box: every rank drives cuda:0 (gloo only; RCCL refuses duplicate GPUs)
    backend = os.environ.get("SRK_DIST_BACKEND") or backend
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [begin, end) of n samples for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def merge_buckets(ranges: Sequence[Tuple[int, int]], min_elems: int) -> List[Tuple[int, int, int]]:
    """Group consecutive backward segments (in execution order; each a contiguous slice, slices descending
    in address) into buckets of at least `min_elems` elements.  Returns (last_segment_index, begin, end)."""
    out: List[Tuple[int, int, int]] = []
    cur_b = cur_e = None
    prev_b = None
    for i, (b, e) in enumerate(ranges):
        if prev_b is not None and e != prev_b:
            raise ValueError("segments are not contiguous/descending")
        prev_b = b
        if cur_b is None:
            cur_b, cur_e = b, e
        else:
            cur_b = b
        if cur_e - cur_b >= min_elems or i == len(ranges) - 1:
            out.append((i, cur_b, cur_e))
            cur_b = cur_e = None
    return out


class GradSynchronizer:
    """Bucketed, overlapped sum-all-reduce of a flat gradient buffer."""

    def __init__(self, segment_ranges: Sequence[Tuple[int, int]], min_bucket_elems: int = 1 << 20,
                 group: Optional[dist.ProcessGroup] = None):
        self.buckets = merge_buckets(segment_ranges, min_bucket_elems)
        self._by_last = {last: (b, e) for last, b, e in self.buckets}
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._side: Optional[torch.cuda.Stream] = None
        self._works = []
        self._flat: Optional[torch.Tensor] = None
        self.time_exposed = False          # bench.py: bracket finish() with events on the compute stream
        self._exposed = []                 # (event before the joins, event after) per finish()

    def bind(self, flat_grad: torch.Tensor) -> None:
        self._flat = flat_grad
        if flat_grad.is_cuda and self._side is None:
            self._side = torch.cuda.Stream(device=flat_grad.device)

    def segment_done(self, seg: int, begin: int = 0, end: int = 0) -> None:
        """Call right after backward segment `seg` has been enqueued on the current stream."""
        if self.world == 1 or seg not in self._by_last:
            return
        b, e = self._by_last[seg]
        chunk = self._flat[b:e]
        if chunk.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """Make the current stream wait for every outstanding bucket."""
        timed = self.time_exposed and self._side is not None and self.world > 1
        if timed:     # the compute stream reaches ev0 when its last backward kernel is done and ev1 when the last bucket is
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream())
        for w in self._works:
            w.wait()
        self._works.clear()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if timed:
            ev1.record(torch.cuda.current_stream())
            self._exposed.append((ev0, ev1))

    def exposed_ms(self) -> Optional[float]:
        """Mean time per step the compute stream spent waiting for all-reduce buckets after its last backward kernel
        (call after a device synchronize); None when nothing was timed."""
        if not self._exposed:
            return None
        total = sum(a.elapsed_time(b) for a, b in self._exposed)
        n = len(self._exposed)
        self._exposed.clear()
        return total / n


class ListGradSynchronizer:
    """Overlapped sum-all-reduce for a model whose backward hands over its parameter gradients as LISTS of tensors, one list per
    backward segment (HAT: tail, RHAG L-1 .. 0, head -- hat_train.hat_backward's ``hook``).  A segment's tensors are flattened into
    one bucket on a side stream (event-ordered behind the segment's kernels), all-reduced asynchronously while the next segment's
    backward kernels run, and copied back (divided by the world size) in ``finish()``, which the backward calls before it returns
    the gradients to autograd.  Device-agnostic (CPU + gloo for the tests)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, average: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.average = average
        self._side: Optional[torch.cuda.Stream] = None
        self._pending = []            # (flat bucket, tensors, work)
        self.buckets_last_step = 0
        self.time_exposed = False
        self._exposed = []

    def segment_done(self, tensors: Sequence[torch.Tensor]) -> None:
        tensors = [t for t in tensors if t is not None]
        if self.world == 1 or not tensors:
            return
        if tensors[0].is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=tensors[0].device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                flat = torch.cat([t.reshape(-1) for t in tensors])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat = torch.cat([t.reshape(-1) for t in tensors])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((flat, tensors, work))

    def finish(self) -> None:
        timed = self.time_exposed and self._side is not None and self.world > 1
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream())
        for _, _, w in self._pending:
            w.wait()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        if timed:
            ev1.record(torch.cuda.current_stream())
            self._exposed.append((ev0, ev1))
        scale = 1.0 / self.world if self.average else 1.0
        for flat, tensors, _ in self._pending:
            off = 0
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                if scale != 1.0:
                    t.mul_(scale)
                off += n
        self.buckets_last_step = len(self._pending)
        self._pending.clear()

    def exposed_ms(self) -> Optional[float]:
        if not self._exposed:
            return None
        total = sum(a.elapsed_time(b) for a, b in self._exposed)
        n = len(self._exposed)
        self._exposed.clear()
        return total / n


class DataParallelSwinIR:
    """Wraps a SwinIR: identical initial weights on every rank, overlapped gradient all-reduce.

        model = SwinIR(...).cuda(); dp = DataParallelSwinIR(model); opt = FusedAdamW(model, grad_div=dp.world)
        out = model(lr_shard); loss.backward(); dp.finish(); opt.step()
    """

    def __init__(self, model, min_bucket_elems: int = 1 << 20, group: Optional[dist.ProcessGroup] = None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._sync: Optional[GradSynchronizer] = None
        self._min_bucket = min_bucket_elems

    def attach(self, device: torch.device) -> None:
        eng = self.model._bind(device)
        if self.world > 1:
            dist.broadcast(eng.flat, src=0, group=self.group)       # same weights on every rank
            eng.packed_valid = False
        self._sync = GradSynchronizer(eng.plan.segment_ranges, self._min_bucket, self.group)
        self._sync.bind(eng.ensure_grad())
        eng.segment_hook = self._sync.segment_done

    def finish(self) -> None:
        if self._sync is not None:
            self._sync.finish()
