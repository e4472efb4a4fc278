#!/bin/bash
tag=${1:-r03f}; LIBS=${2:-"cur.so noview.so noview_nopipe.so occ5.so"}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
export CLIPPER_HIP_ROW_VIEW=0
for round in 1 2; do
  for m in 10000 30000; do
    (cd tools/_bin/r02tree && timeout 60 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('r02 $m', d['value'], d['solve_ms'], d['gemv_avg_us'])" | tee -a $out/ab.txt
    for lib in $LIBS; do
      CLIPPER_HIP_LIB=$PWD/clipper_amd/lib/variants/$lib timeout 60 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib $m', d['value'], d['solve_ms'], d['gemv_avg_us'])" | tee -a $out/ab.txt
    done
  done
done
