set -e
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_dat.py -q -k "train or rect or errors" > gpurun_out/r3m_tests.log 2>&1 || true
tail -40 gpurun_out/r3m_tests.log
