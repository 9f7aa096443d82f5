"""Developer sweep of the streaming-GEMM tile shape: run cfg3 train steps under (rows per tile, K split)
overrides; the kernel names in a rocprofv3 kernel trace carry the template arguments, so one trace gives the
per-configuration duration of every block GEMM:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep -- python3 tools/stream_sweep.py [steps] [sweep|split|default]
    python tools/stream_sweep.py --summary gpurun_out/sweep
"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {0: "BF16", 1: "QKV", 2: "PROJ_RES", 3: "GELU", 4: "RES", 5: "DGELU", 13: "LNBWD"}


def summary(path):
    f = glob.glob(os.path.join(path, "*", "*_kernel_stats.csv"))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        if "gemm_stream" in r["Name"]:
            a = re.search(r"<(.*)>", r["Name"]).group(1).split(", ")
            kind = ("split" + a[3]) if "split_kernel" in r["Name"] else ("ks2" if a[3] == "true" else "sym")
            rows.append((NAMES.get(int(a[0]), a[0]), 64 * int(a[1]), int(a[2]), kind, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    for r in sorted(rows):
        print(f"{r[0]:9s} K={r[1]:3d} BM={r[2]:2d} {r[3]:12s} calls={r[4]:4d} avg={r[5]:7.1f} us")


def main():
    import torch
    from oracle import swinir_oracle as O
    import tpu_superresolution_amd as T
    from tpu_superresolution_amd._lib import check, lib

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mode = sys.argv[2] if len(sys.argv) > 2 else "sweep"
    cfg = O.SwinIRConfig.classical_x4()
    m = T.SwinIR(**cfg.kwargs()).cuda().train()
    x = torch.rand(32, 3, 64, 64, device="cuda")
    t = torch.rand(32, 3, 256, 256, device="cuda")
    if mode == "sweep":
        combos = [(bm, ks2, 0) for bm in (16, 32, 64) for ks2 in (0, 1)] + [(bm, -1, 1) for bm in (16, 32, 64)]
    elif mode == "split":
        combos = [(bm, -1, 1) for bm in (16, 32, 64)]
    elif mode == "nb8":
        combos = [(32, -1, 1, 8), (64, -1, 1, 8)]
    else:
        combos = [(0, -1, -1)]
    for combo in combos:
        bm, ks2, split = combo[:3]
        check(lib().srk_set_option(b"gemm_stream_nb", combo[3] if len(combo) > 3 else 0))
        check(lib().srk_set_option(b"gemm_stream_bm", bm))
        check(lib().srk_set_option(b"gemm_stream_ks2", ks2))
        check(lib().srk_set_option(b"gemm_stream_split", split))
        for _ in range(steps):
            loss = torch.nn.functional.l1_loss(m(x), t)
            loss.backward()
            for p in m.parameters():
                p.grad = None
        torch.cuda.synchronize()
        print(f"bm={bm} ks2={ks2} split={split}: loss {float(loss.detach()):.5f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summary":
        summary(sys.argv[2])
    else:
        main()
