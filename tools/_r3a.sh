set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_attn_bwd_fused.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/a_tests1.log 2>&1 || { tail -30 gpurun_out/a_tests1.log; exit 1; }
tail -2 gpurun_out/a_tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_emulation.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/a_tests2.log 2>&1 || { tail -30 gpurun_out/a_tests2.log; exit 1; }
tail -2 gpurun_out/a_tests2.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --opt attn_bwd_fused=0 > gpurun_out/a_bench_off.json 2> gpurun_out/a_bench_off.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/a_bench_on.json 2> gpurun_out/a_bench_on.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --opt attn_bwd_fused=0 > gpurun_out/a_bench_off2.json 2> gpurun_out/a_bench_off2.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/a_bench_on2.json 2> gpurun_out/a_bench_on2.err
python - <<'PY'
import json
for n in ("off","on","off2","on2"):
    d = json.loads(open(f"gpurun_out/a_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],3))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/a_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/a_prof.err
cd /root/repo
python tools/prof_summary.py gpurun_out/a_prof 13
