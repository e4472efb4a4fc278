#!/bin/bash
# window size / workgroup-count sweep of the compressed-copy solver. usage: tools/gpu_csc_sweep.sh <tag> "<V list>" "<wgs list>" "<m list>"
TAG=$1; VL=${2:-"6"}; WL=${3:-"768"}; ML=${4:-"10000"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for m in $ML; do for V in $VL; do for W in $WL; do
  CLIPPER_HIP_WINDOW=$V CLIPPER_HIP_CSC_WGS=$W timeout 600 python bench.py --storage csc --m $m --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('m', d['config']['m'], 'V=$V wgs=$W', 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'solve', d['solve_ms'], 'passes', d['gemv_passes_per_solve'], 'gemv_us', d['gemv_avg_us'], 'bytes', d['roofline']['bytes_per_launch'])" 2>&1 | tee -a $OUT/sweep.log
done; done; done
