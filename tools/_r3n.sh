set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_dat.py -q -k "train or rect" > gpurun_out/r3n_tests.log 2>&1 || true
tail -5 gpurun_out/r3n_tests.log
timeout -k 10 400 python bench.py --config cfg5 --train > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err || (tail -20 gpurun_out/r3n_bench.err; false)
cut -c1-400 gpurun_out/r3n_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3n_prof -- python3 /root/repo/bench.py --config cfg5 --train --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/r3n_prof.err
