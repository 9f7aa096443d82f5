# developer: SQ counters per kernel for a HAT / DAT train step   usage: bash tools/_sq_cfg.sh cfg4|cfg5
set -e
CFG=${1:-cfg4}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/q_prof /root/repo/gpurun_out/q_pmcA /root/repo/gpurun_out/q_pmcB
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/q_prof -- python3 /root/repo/bench.py --config $CFG --train --no-graph --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/q_prof.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /root/repo/gpurun_out/q_pmcA -- python3 /root/repo/bench.py --config $CFG --train --no-graph --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/q_pmcA.err
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d /root/repo/gpurun_out/q_pmcB -- python3 /root/repo/bench.py --config $CFG --train --no-graph --steps 2 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/q_pmcB.err
cd /root/repo
python tools/sq_summary.py gpurun_out q_ > gpurun_out/q_sq_$CFG.txt
head -16 gpurun_out/q_sq_$CFG.txt
