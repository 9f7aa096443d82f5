set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_attn_bwd_fused.py tests/test_gpu_hat.py -x -q -m gpu > gpurun_out/f_tests1.log 2>&1 || { tail -30 gpurun_out/f_tests1.log; exit 1; }
tail -2 gpurun_out/f_tests1.log
SRK_LIB_PATH=$PWD/tpu_superresolution_amd/_variants/abf_probe.so timeout -k 10 200 python tools/abf_bench.py 3 --probe > gpurun_out/f_probe.txt 2>&1 || { tail -20 gpurun_out/f_probe.txt; exit 1; }
head -20 gpurun_out/f_probe.txt | cut -c1-120
grep -A18 "wave 11" gpurun_out/f_probe.txt | cut -c1-120
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/f_bench_on.json 2> gpurun_out/f_bench_on.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --opt attn_bwd_fused=0 > gpurun_out/f_bench_off.json 2> gpurun_out/f_bench_off.err
python -c "
import json
for n in ('on','off'):
    d = json.loads(open('gpurun_out/f_bench_%s.json' % n).read().strip().splitlines()[-1]); print('bench', n, round(d['ms_per_step'],3), d['config'].get('ms_per_step_median'))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/f_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/f_prof.err
cd /root/repo
python tools/prof_summary.py gpurun_out/f_prof 13 | head -14
timeout -k 10 600 python bench.py --config cfg4 --train --steps 5 --warmup 2 > gpurun_out/f_bench_cfg4_train.json 2> gpurun_out/f_bench_cfg4_train.err
python -c "
import json
d = json.loads(open('gpurun_out/f_bench_cfg4_train.json').read().strip().splitlines()[-1]); print('cfg4 train', round(d['ms_per_step'],2))"
