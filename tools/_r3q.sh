set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py tests/test_gpu_hat.py -q -x > gpurun_out/r3q_tests.log 2>&1 || (tail -30 gpurun_out/r3q_tests.log; false)
tail -2 gpurun_out/r3q_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/r3q_prof /root/repo/gpurun_out/r3q_fetch
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3q_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /root/repo/gpurun_out/r3q_bench.json 2>/root/repo/gpurun_out/r3q_prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r3q_fetch -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/r3q_fetch.err
cut -c1-300 /root/repo/gpurun_out/r3q_bench.json
