set -e
cd /root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3r_tests.log 2>&1 || (tail -40 gpurun_out/r3r_tests.log; false)
tail -2 gpurun_out/r3r_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err
cut -c1-330 gpurun_out/r3r_bench.json
