set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_model.py tests/test_gpu_emulation.py tests/test_gpu_stream_gemm.py -q -x > gpurun_out/r3o_tests.log 2>&1 || (tail -30 gpurun_out/r3o_tests.log; false)
tail -3 gpurun_out/r3o_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3o_bench.json 2> gpurun_out/r3o_bench.err
cut -c1-330 gpurun_out/r3o_bench.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r3o_bench2.json 2> gpurun_out/r3o_bench2.err
cut -c1-330 gpurun_out/r3o_bench2.json
