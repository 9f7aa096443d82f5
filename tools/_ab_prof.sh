# developer A/B: kernel stats of the default bench with an option on / off   (usage: bash tools/_ab_prof.sh <option>)
set -e
OPT=$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > /dev/null 2>/root/repo/gpurun_out/ab1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab0 -- python3 /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --opt $OPT=0 > /dev/null 2>/root/repo/gpurun_out/ab0.err
cd /root/repo
for d in ab1 ab0; do
  f=$(find gpurun_out/$d -name '*kernel_stats.csv' | head -1)
  echo "== $d"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'])
PY
done
