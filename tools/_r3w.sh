set -e
cd /root/repo
timeout -k 10 400 python -m pytest tests/test_gpu_hat.py -q > gpurun_out/r3w_tests.log 2>&1 || (tail -40 gpurun_out/r3w_tests.log; false)
tail -2 gpurun_out/r3w_tests.log
timeout -k 10 400 python bench.py --config cfg4 --train > gpurun_out/r3w_cfg4t.json 2> gpurun_out/r3w_cfg4t.err || (tail -20 gpurun_out/r3w_cfg4t.err; false)
cut -c1-260 gpurun_out/r3w_cfg4t.json
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/r3w_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3w_prof -- python3 /root/repo/bench.py --config cfg4 --train --no-graph --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/r3w.err
