set -e
cd /tmp && export TMPDIR=/tmp
for n in 0 3 5 8 12; do
  rm -rf /root/repo/gpurun_out/r3s_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3s_$n -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 --opt conv_stagger=$n > /root/repo/gpurun_out/r3s_$n.json 2>/root/repo/gpurun_out/r3s_$n.err
  echo "stagger $n: $(cut -c150-200 /root/repo/gpurun_out/r3s_$n.json)"
done
