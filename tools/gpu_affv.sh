#!/bin/bash
for round in 1 2; do for lib in at4.so at8.so; do for m in 10000 30000; do
  CLIPPER_HIP_LIB=$PWD/clipper_amd/lib/variants/$lib timeout 300 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$lib', d['config']['m'], 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'aff_kernel', d['affinity_kernel_ms'], 'solve', d['solve_ms'], 'score', d['solution']['score'])"
done; done; done
