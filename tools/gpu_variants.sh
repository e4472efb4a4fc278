#!/bin/bash
# A/B of library variants in ONE box. usage: tools/gpu_variants.sh <tag> "<variant .so list>" "<V list>" "<m list>" [rounds]
TAG=$1; LIBS=$2; VL=${3:-"6"}; ML=${4:-"10000"}; ROUNDS=${5:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in $(seq $ROUNDS); do for lib in $LIBS; do for V in $VL; do for m in $ML; do
  CLIPPER_HIP_LIB=$PWD/clipper_amd/lib/variants/$lib CLIPPER_HIP_WINDOW=$V timeout 300 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$lib V=$V', d['config']['m'], 'step', d['ms_per_step'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'per_pass_us', round(1e3*d['solve_ms']/d['gemv_passes_per_solve'],1))" >> $OUT/ab.log 2>&1
done; done; done; done
cat $OUT/ab.log
