set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_hat.py -x -q -m gpu -k "use_checkpoint or hat" > gpurun_out/j_tests1.log 2>&1 || { tail -40 gpurun_out/j_tests1.log; exit 1; }
tail -2 gpurun_out/j_tests1.log
timeout -k 10 600 python bench.py --config cfg4 --train --steps 5 --warmup 2 > gpurun_out/j_bench_cfg4_train.json 2> gpurun_out/j_bench_cfg4_train.err
python -c "
import json
d = json.loads(open('gpurun_out/j_bench_cfg4_train.json').read().strip().splitlines()[-1]); print('cfg4 train', round(d['ms_per_step'],2))"
