set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/r3u_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3u_prof -- python3 /root/repo/bench.py --config cfg5 --train --steps 3 --warmup 1 > /root/repo/gpurun_out/r3u.json 2>/root/repo/gpurun_out/r3u.err
cut -c1-260 /root/repo/gpurun_out/r3u.json
