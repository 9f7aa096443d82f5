set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_dat.py tests/test_gpu_hat.py -x -q -m gpu > gpurun_out/k_tests1.log 2>&1 || { tail -40 gpurun_out/k_tests1.log; exit 1; }
tail -2 gpurun_out/k_tests1.log
