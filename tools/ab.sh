#!/bin/bash
# Developer A/B: bench every library variant under tpu_superresolution_amd/_variants/ (built with SRK_EXTRA_FLAGS), alternating.
set -e
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for v in tpu_superresolution_amd/_variants/*.so; do
    ms=$(SRK_LIB_PATH=$PWD/$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'])")
    echo "round $round $(basename $v) $ms"
  done
done
