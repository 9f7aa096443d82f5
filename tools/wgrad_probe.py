"""Developer probe: the streaming linear weight-gradient kernel on its own (outside a train step), HIP-event timed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpu_superresolution_amd._lib import check, lib

def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

st = torch.cuda.current_stream().cuda_stream
for M, N, K in ((65536, 192, 192), (131072, 192, 192), (262144, 192, 192), (1048576, 192, 192), (524288, 192, 192), (131072 * 3, 576, 192), (131072 * 2, 192, 384)):
    for fill, partials in (("randn", 0), ("randn", 1)):
        check(lib().srk_set_option(b"wgrad_partials", partials))
        mk = (lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16)) if fill == "zeros" else \
             (lambda *s: torch.randn(*s, device="cuda", dtype=torch.bfloat16))
        ys = [mk(M, N) for _ in range(2)]
        xs = [mk(M, K) for _ in range(2)]
        dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
        k = [0]
        def run():
            i = k[0] & 1; k[0] += 1
            check(lib().srk_linear_wgrad_bf16(ys[i].data_ptr(), xs[i].data_ptr(), dw.data_ptr(), db.data_ptr(), M, N, K, st))
        t = timeit(run)
        mb = M * (N + K) * 2 / 1e6
        print(f"wgrad M {M:8d} N {N} K {K} partials {partials}: {t:8.1f} us   {mb:7.1f} MB unique  {mb / t:5.2f} TB/s", flush=True)
        del ys, xs
