#!/bin/bash
# GPU suite + A/B of the line-search window sizes. usage: tools/gpu_ab.sh <tag> [sizes...]
TAG=${1:-ab}; shift; SIZES=${@:-"1000 10000"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $OUT/pytest.log
for V in 1 4 6 8; do
  for m in $SIZES; do
    CLIPPER_HIP_WINDOW=$V timeout 300 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('V=$V', d['config']['m'], 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'frac', d['roofline']['frac'], 'score', d['solution']['score'])" >> $OUT/ab.log 2>&1
  done
done
tail -8 $OUT/pytest.log; cat $OUT/ab.log
