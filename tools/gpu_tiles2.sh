#!/bin/bash
for t in 5 4 8 9 13 17; do
  CLIPPER_HIP_TILES=$t timeout 300 python bench.py --m 30000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('tiles=$t', d['config']['m'], 'step', d['ms_per_step'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'min', d['gemv_min_us'])"
done
