set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "fused_backward or deterministic" > gpurun_out/h_tests1.log 2>&1 || { tail -40 gpurun_out/h_tests1.log; exit 1; }
tail -2 gpurun_out/h_tests1.log
for v in 1 0 1 0; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --opt mlp_bwd_fused=$v > gpurun_out/h_bench_$v.json 2> gpurun_out/h_bench_$v.err
python -c "
import json
d = json.loads(open('gpurun_out/h_bench_$v.json').read().strip().splitlines()[-1]); print('bench mlp_bwd_fused=$v', round(d['ms_per_step'],3), d['config'].get('ms_per_step_median'))"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/h_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/h_prof.err
cd /root/repo
python tools/prof_summary.py gpurun_out/h_prof 13 | head -12
