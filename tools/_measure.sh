# Round-end evidence run (GPU box): full GPU test suite, bench lines, rocprofv3 kernel stats, PMC traffic and SQ counters.
# Everything lands in gpurun_out/m_*; tools/install_profiles.py <tag> copies the summaries into profiles/.
set -e
cd /root/repo
TAG=${1:-r03}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/m_gpu_tests.log 2>&1
tail -2 gpurun_out/m_gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
timeout -k 10 300 python bench.py --config cfg2 > gpurun_out/m_bench_cfg2.json 2> gpurun_out/m_bench_cfg2.err
timeout -k 10 300 python bench.py --config cfg4 > gpurun_out/m_bench_cfg4.json 2> gpurun_out/m_bench_cfg4.err
timeout -k 10 300 python bench.py --config cfg5 > gpurun_out/m_bench_cfg5.json 2> gpurun_out/m_bench_cfg5.err
timeout -k 10 400 python bench.py --config cfg4 --train > gpurun_out/m_bench_cfg4_train.json 2> gpurun_out/m_bench_cfg4_train.err
timeout -k 10 400 python bench.py --config cfg5 --train > gpurun_out/m_bench_cfg5_train.json 2> gpurun_out/m_bench_cfg5_train.err
timeout -k 10 300 python bench.py --use-checkpoint --no-cpu-baseline --no-roofline > gpurun_out/m_bench_cfg3_use_checkpoint.json 2> gpurun_out/m_bench_cfg3_use_checkpoint.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof -- python3 /root/repo/bench.py --no-cpu-baseline --steps 20 > /root/repo/gpurun_out/m_prof_bench.json 2>/root/repo/gpurun_out/m_prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/m_pmc_fetch -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/m_pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/m_pmc_write -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/m_pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/s_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/s_prof.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /root/repo/gpurun_out/s_pmcA -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/s_pmcA.err
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d /root/repo/gpurun_out/s_pmcB -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/s_pmcB.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg4 -- python3 /root/repo/bench.py --config cfg4 --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 2 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg4.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg4_train -- python3 /root/repo/bench.py --config cfg4 --train --no-graph --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg4_train.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg5_train -- python3 /root/repo/bench.py --config cfg5 --train --no-graph --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg5_train.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg5 -- python3 /root/repo/bench.py --config cfg5 --no-cpu-baseline --no-roofline --no-graph --steps 10 --warmup 2 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/m_prof_cfg2 -- python3 /root/repo/bench.py --config cfg2 --no-cpu-baseline --no-roofline --no-graph --steps 50 --warmup 2 > /dev/null 2>/root/repo/gpurun_out/m_prof_cfg2.err
cd /root/repo
python tools/sq_summary.py gpurun_out s_ > gpurun_out/m_sq.txt
timeout -k 10 120 python __graft_entry__.py --smoke > gpurun_out/m_smoke.log 2>&1
tail -2 gpurun_out/m_smoke.log
cut -c1-500 gpurun_out/m_bench.json
