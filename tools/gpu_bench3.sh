#!/bin/bash
# three default bench runs (no CPU baseline), one line each. usage: tools/gpu_bench3.sh <tag> [extra bench args]
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2 3; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['config']['storage'], 'm', d['config']['m'], 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'affk', d['affinity_kernel_ms'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'frac', d['roofline']['frac'])" 2>&1 | tee -a $OUT/bench3.log; done
