set -e
cd /root/repo
timeout -k 10 600 python bench.py --config cfg4 --train --steps 5 --warmup 2 > gpurun_out/e_bench_cfg4_train.json 2> gpurun_out/e_bench_cfg4_train.err || { tail -30 gpurun_out/e_bench_cfg4_train.err; exit 1; }
cut -c1-600 gpurun_out/e_bench_cfg4_train.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/e_prof -- python3 /root/repo/bench.py --config cfg4 --train --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/e_prof.err
cd /root/repo
python tools/prof_summary.py gpurun_out/e_prof 4 | head -40
