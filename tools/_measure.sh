set -e
cd /root/repo
TAG=${1:-r02}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/m_gpu_tests.log 2>&1
tail -2 gpurun_out/m_gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
timeout -k 10 300 python bench.py --config cfg2 > gpurun_out/m_bench_cfg2.json 2> gpurun_out/m_bench_cfg2.err
timeout -k 10 300 python bench.py --config cfg4 > gpurun_out/m_bench_cfg4.json 2> gpurun_out/m_bench_cfg4.err
timeout -k 10 300 python bench.py --config cfg5 > gpurun_out/m_bench_cfg5.json 2> gpurun_out/m_bench_cfg5.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof -- python3 /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/m_prof_bench.json 2>/root/repo/gpurun_out/m_prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/m_pmc_fetch -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/m_pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/m_pmc_write -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/m_pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg4 -- python3 /root/repo/bench.py --config cfg4 --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 2 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg4.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg5 -- python3 /root/repo/bench.py --config cfg5 --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 2 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg2 -- python3 /root/repo/bench.py --config cfg2 --no-cpu-baseline --no-roofline --no-graph --steps 50 --warmup 2 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg2.err
cd /root/repo
cut -c1-400 gpurun_out/m_bench.json
