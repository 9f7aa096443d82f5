set -e
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/s_prof -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/s_prof.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /root/repo/gpurun_out/s_pmcA -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/s_pmcA.err
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d /root/repo/gpurun_out/s_pmcB -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/s_pmcB.err
cd /root/repo
python tools/prof_summary.py gpurun_out/s_prof 30
