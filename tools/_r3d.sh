set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_hat.py -x -q -m gpu > gpurun_out/d_tests.log 2>&1 || { tail -60 gpurun_out/d_tests.log; exit 1; }
tail -3 gpurun_out/d_tests.log
