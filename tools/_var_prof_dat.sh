# developer A/B: per-kernel averages of the DAT train step (eager) for every library under tpu_superresolution_amd/_variants/
set -e
cd /tmp && export TMPDIR=/tmp
for so in /root/repo/tpu_superresolution_amd/_variants/*.so; do
  v=$(basename $so .so)
  rm -rf /root/repo/gpurun_out/vp_$v
  SRK_LIB_PATH=$so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/vp_$v -- python3 /root/repo/bench.py --config ${2:-cfg5} --train --no-graph --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/vp_$v.err
  f=$(find /root/repo/gpurun_out/vp_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python3 - "$f" "${1:-attn}" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step", tot/4/1e6)
for r in rows[:40]:
    if sys.argv[2] in r['Name'] : print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'])
PY
done
