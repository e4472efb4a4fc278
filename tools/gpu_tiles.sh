#!/bin/bash
OUT=gpurun_out/tiles; mkdir -p $OUT
for round in 1 2; do for t in 13 12 11 10 8 6 16 19 25; do
  CLIPPER_HIP_TILES=$t CLIPPER_HIP_WINDOW=6 timeout 300 python bench.py --m ${1:-10000} --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('tiles=$t', d['config']['m'], 'step', d['ms_per_step'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'min', d['gemv_min_us'])" >> $OUT/ab.log 2>&1
done; done
cat $OUT/ab.log
