"""The data-parallel training path driven by the real engine on the GPU: two ranks (separate processes) share cuda:0 and
all-reduce over gloo -- a one-GPU box cannot host two RCCL ranks -- which exercises everything except the RCCL transport
itself: weight broadcast in attach(), the per-segment hook of the C backward, the event / side-stream / async all_reduce
branch of GradSynchronizer, the join before the optimizer and FusedAdamW's division by the world size.

Check: the summed gradient of world = 2 over two half batches, divided by 2, equals the single-process gradient
of the whole batch (mean-reduced loss), and so do the post-step weights, up to the bf16 noise of the activation-gradient
streams (the per-shard runs round differently from the whole-batch run: same bound as the batch-split test of
test_gpu_full_size.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_match_the_single_process_step(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dp_rehearsal
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.pt")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SRK_SHARE_GPU="1", SRK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dp_rehearsal.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = torch.load(out, weights_only=True)
    assert got["bad"] == 0 and got["buckets"] >= 2           # more than one bucket: the overlap path really ran
    # single process, whole batch
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd.optim import FusedAdamW
    from tpu_superresolution_amd.training import train_step
    cfg, sd, x, t = dp_rehearsal.rehearsal_case()
    m = T.SwinIR(drop_path_rate=0.0, **cfg.kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
    loss, _ = train_step(m, opt, x.cuda(), t.cuda())
    eng = m._engine
    g1, w1 = eng.flat_grad.cpu(), eng.flat.cpu()
    g2 = got["grad"] / 2.0                                    # ranks sum; the optimizer divides by the world size
    rel = float((g2 - g1).norm() / g1.norm())
    worst = 0.0
    for p in eng.plan.params:
        a, b = g2[p.offset:p.offset + p.numel], g1[p.offset:p.offset + p.numel]
        worst = max(worst, float((a - b).norm() / (b.norm() + 1e-20)))
    print(f"world-2 vs world-1: flat gradient rel-L2 {rel:.3e}, worst tensor {worst:.3e}")
    assert rel <= 5e-3 and worst <= 1.5e-2
    # the two half-batch means average to the whole-batch mean loss; rank 0 reports its own shard's loss
    assert abs(got["loss"] - float(loss)) <= 0.05 * float(loss)
    # post-step weights: Adam's first step moves every weight by ~lr * sign(g); compare where the gradient is not ~0
    moved = (got["flat"] - w1).abs()
    assert float(moved.max()) <= 2.1e-3                       # at most 2 * lr apart (opposite signs on near-zero gradients)
    assert float((moved > 1e-4).float().mean()) <= 0.02
