set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_hat.py tests/test_gpu_dat.py tests/test_abi_and_host.py -q > gpurun_out/r3x_tests.log 2>&1 || (tail -40 gpurun_out/r3x_tests.log; false)
tail -2 gpurun_out/r3x_tests.log
timeout -k 10 400 python bench.py --config cfg4 --train > gpurun_out/r3x_cfg4t.json 2> gpurun_out/r3x_cfg4t.err
cut -c1-240 gpurun_out/r3x_cfg4t.json
timeout -k 10 400 python bench.py --config cfg5 --train > gpurun_out/r3x_cfg5t.json 2> gpurun_out/r3x_cfg5t.err
cut -c1-240 gpurun_out/r3x_cfg5t.json
