set -e
cd /root/repo
timeout -k 10 400 python -m pytest tests/test_gpu_hat.py -q -s -k "fused_mlp or gradients or learns" > gpurun_out/r3w_tests.log 2>&1 || (tail -40 gpurun_out/r3w_tests.log; false)
tail -4 gpurun_out/r3w_tests.log
timeout -k 10 400 python bench.py --config cfg4 --train > gpurun_out/r3w_cfg4t.json 2> gpurun_out/r3w_cfg4t.err || (tail -20 gpurun_out/r3w_cfg4t.err; false)
cut -c1-260 gpurun_out/r3w_cfg4t.json
