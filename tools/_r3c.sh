set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_attn_bwd_fused.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/c_tests1.log 2>&1 || { tail -30 gpurun_out/c_tests1.log; exit 1; }
tail -2 gpurun_out/c_tests1.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/c_bench_on.json 2> gpurun_out/c_bench_on.err
python -c "
import json
d = json.loads(open('gpurun_out/c_bench_on.json').read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'],3), d['config'].get('ms_per_step_median'))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/c_prof.err
cd /root/repo
python tools/prof_summary.py gpurun_out/c_prof 13 | head -16
