#!/bin/bash
# profile on/off A/B. usage: tools/gpu_quick2.sh <tag> "<V list>" "<m list>"
TAG=${1:-q}; VL=${2:-"1 6"}; ML=${3:-"1000 10000"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for extra in "" "--no-profile"; do for V in $VL; do for m in $ML; do
  CLIPPER_HIP_WINDOW=$V timeout 300 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline $extra 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('V=$V $extra', d['config']['m'], 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'])" >> $OUT/ab.log 2>&1
done; done; done
cat $OUT/ab.log
