set -e
cd /root/repo
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/l_tests.log 2>&1 || { tail -40 gpurun_out/l_tests.log; exit 1; }
tail -3 gpurun_out/l_tests.log
