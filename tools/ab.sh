#!/bin/bash
# Developer A/B: bench every library variant under tpu_superresolution_amd/_variants/ (built with SRK_EXTRA_FLAGS), alternating.
set -e
cd "$(dirname "$0")/.."
# the first run after a pause is a little slower (clocks): one throw-away run, then alternate the order every round
SRK_LIB_PATH=$PWD/$(ls tpu_superresolution_amd/_variants/*.so | head -1) timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 >/dev/null 2>&1 || true
for round in 1 2 3 4; do
  if [ $((round % 2)) -eq 1 ]; then order=$(ls tpu_superresolution_amd/_variants/*.so); else order=$(ls -r tpu_superresolution_amd/_variants/*.so); fi
  for v in $order; do
    ms=$(SRK_LIB_PATH=$PWD/$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'])")
    echo "round $round $(basename $v) $ms"
  done
done
