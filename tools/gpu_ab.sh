#!/bin/bash
# A/B of the launch shapes at several sizes. usage: tools/gpu_ab.sh <tag>
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
for mode in legacy split fused; do
  for m in 1000 10000; do
    CLIPPER_HIP_PASS=$mode timeout 300 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$mode', d['config']['m'], 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'frac', d['roofline']['frac'], 'score', d['solution']['score'])" >> $OUT/ab.log 2>&1
  done
done
tail -5 $OUT/pytest.log; cat $OUT/ab.log
