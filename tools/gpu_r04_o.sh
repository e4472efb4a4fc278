#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04o
for i in 1 2 3; do
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --probe-m 0 > gpurun_out/r04o/bench_$i.log 2>&1
  grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*\|"affinity_ms": [0-9.]*' gpurun_out/r04o/bench_$i.log | head -3 | tr '\n' ' '; echo
done
CLIPPER_HIP_VIEW_RESIDENT=0 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --probe-m 0 > gpurun_out/r04o/bench_streamed.log 2>&1
echo streamed; grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*' gpurun_out/r04o/bench_streamed.log | head -2 | tr '\n' ' '; echo
CLIPPER_HIP_ROW_VIEW=0 timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --probe-m 0 > gpurun_out/r04o/bench_noviews.log 2>&1
echo noviews; grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*' gpurun_out/r04o/bench_noviews.log | head -2 | tr '\n' ' '; echo
CLIPPER_HIP_STAMPS=1 timeout 120 python tools/rvr_timeline.py 2>&1 | grep -v Warn | head -12
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile 2>&1 >/dev/null | tail -6
