set -e
cd /root/repo
timeout -k 10 400 python -m pytest tests/test_gpu_dat.py -q > gpurun_out/r3n_tests.log 2>&1 || (tail -30 gpurun_out/r3n_tests.log; false)
tail -2 gpurun_out/r3n_tests.log
timeout -k 10 400 python bench.py --config cfg5 --train > gpurun_out/r3n_bench.json 2> gpurun_out/r3n_bench.err || (tail -20 gpurun_out/r3n_bench.err; false)
cut -c1-300 gpurun_out/r3n_bench.json
timeout -k 10 400 python bench.py --config cfg5 > gpurun_out/r3n_bench_inf.json 2> gpurun_out/r3n_bench_inf.err
cut -c1-300 gpurun_out/r3n_bench_inf.json
