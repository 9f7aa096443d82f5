set -e
cd /root/repo
timeout -k 10 200 python tools/abf_bench.py 30
SRK_LIB_PATH=$PWD/tpu_superresolution_amd/_variants/abf_probe.so timeout -k 10 200 python tools/abf_bench.py 3 --probe > gpurun_out/b_probe.txt 2>&1 || { tail -20 gpurun_out/b_probe.txt; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/b_bench_on.json 2> gpurun_out/b_bench_on.err
python -c "
import json
d = json.loads(open('gpurun_out/b_bench_on.json').read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'],3), d['config'].get('ms_per_step_median'))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/k_prof -- python3 /root/repo/tools/abf_bench.py 10 > /dev/null 2>/root/repo/gpurun_out/k_prof.err
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /root/repo/gpurun_out/k_pmcA -- python3 /root/repo/tools/abf_bench.py 5 > /dev/null 2>/root/repo/gpurun_out/k_pmcA.err
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d /root/repo/gpurun_out/k_pmcB -- python3 /root/repo/tools/abf_bench.py 5 > /dev/null 2>/root/repo/gpurun_out/k_pmcB.err
cd /root/repo
python tools/sq_summary.py gpurun_out k_
