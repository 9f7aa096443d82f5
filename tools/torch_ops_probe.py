"""Developer probe: which torch ops (not libsrk launches) a HAT / DAT forward issues -- they show up as copy / index kernels."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for name in sys.argv[1:] or ["cfg4", "cfg5"]:
    m = bench.build_infer_model(name, torch.device("cuda"))
    x = torch.rand(16, 3, 64, 64, device="cuda")
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            m(x)
            torch.cuda.synchronize()
    print(name)
    print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=14, max_name_column_width=60, max_src_column_width=90))
