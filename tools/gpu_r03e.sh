#!/bin/bash
tag=${1:-r03e}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for i in 1 2 3; do
  (cd tools/_bin/r02tree && timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('r02 ', d['value'], d['affinity_ms'], d['solve_ms'], d['gemv_avg_us'])" | tee -a $out/ab.txt
  timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('now ', d['value'], d['affinity_ms'], d['solve_ms'], d['gemv_avg_us'])" | tee -a $out/ab.txt
  CLIPPER_HIP_ROW_VIEW=0 timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nowV0', d['value'], d['affinity_ms'], d['solve_ms'], d['gemv_avg_us'])" | tee -a $out/ab.txt
done
timeout 150 python tools/rowview_probe.py --m 10000 30000 100000 --reps 3 --profile > $out/probe.jsonl 2>&1
echo "probe rc=$?" | tee -a $out/summary.txt
cat $out/probe.jsonl | cut -c1-1500
timeout 100 python -m pytest tests/test_gpu_rowview.py tests/test_gpu_parity.py -x -q > $out/some_tests.txt 2>&1
echo "tests rc=$?" | tee -a $out/summary.txt
tail -3 $out/some_tests.txt
