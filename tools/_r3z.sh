set -e
cd /root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3z_tests.log 2>&1 || (tail -40 gpurun_out/r3z_tests.log; false)
tail -2 gpurun_out/r3z_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3z_bench.json 2> gpurun_out/r3z_bench.err
cut -c1-300 gpurun_out/r3z_bench.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --opt mlp_bwd_fused=0 --opt attn_bwd_fused=0 > gpurun_out/r3z_bench_off.json 2> gpurun_out/r3z_bench_off.err
cut -c1-300 gpurun_out/r3z_bench_off.json
