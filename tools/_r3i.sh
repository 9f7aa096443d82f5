set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_hat.py -x -q -m gpu > gpurun_out/i_tests1.log 2>&1 || { tail -40 gpurun_out/i_tests1.log; exit 1; }
tail -2 gpurun_out/i_tests1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/i_prof -- python3 /root/repo/bench.py --config cfg4 --train --steps 3 --warmup 1 > /root/repo/gpurun_out/i_bench.json 2>/root/repo/gpurun_out/i_prof.err
cd /root/repo
python tools/prof_summary.py gpurun_out/i_prof 4 | head -16
