# developer: kernel stats of the DAT train step (eager launches)   usage: bash tools/_dat_prof.sh <tag> [cfg]
set -e
TAG=$1; CFG=${2:-cfg5}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/dp_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/dp_$TAG -- python3 /root/repo/bench.py --config $CFG --train --no-graph --steps 3 --warmup 1 > /dev/null 2>/root/repo/gpurun_out/dp_$TAG.err
cd /root/repo
f=$(find gpurun_out/dp_$TAG -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=4
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step', tot/steps/1e6, 'launches/step', sum(int(r['Calls']) for r in rows)/steps)
for r in rows[:26]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls'])/steps:8.1f} {float(r['TotalDurationNs'])/steps/1e6:7.3f} {float(r['AverageNs'])/1e3:8.1f}")
small=sum(float(r['TotalDurationNs']) for r in rows if float(r['AverageNs'])<8000)/steps/1e6
print('kernels <8us: ms/step', small, 'launches/step', sum(int(r['Calls']) for r in rows if float(r['AverageNs'])<8000)/steps)
PY
