set -e
cd /root/repo
timeout -k 10 400 python -m pytest tests/test_gpu_dat.py -q > gpurun_out/r3n_tests.log 2>&1 || (tail -30 gpurun_out/r3n_tests.log; false)
tail -2 gpurun_out/r3n_tests.log
timeout -k 10 400 python bench.py --config cfg5 --train > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err || (tail -20 gpurun_out/r3n_bench.err; false)
cut -c1-220 gpurun_out/r3n_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/r3n_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3n_prof -- python3 /root/repo/bench.py --config cfg5 --train --no-graph --steps 3 --warmup 1 > /root/repo/gpurun_out/r3n.json 2>/root/repo/gpurun_out/r3n.err
