#!/bin/bash
OUT=gpurun_out/aff; mkdir -p $OUT
( timeout 600 python -m pytest tests -m gpu -x -q -k "affinity or golden or parity or pointnormal or dimension or duplicate or smoke" 2>&1 | tail -12 ) > $OUT/pytest.log
tail -4 $OUT/pytest.log
for mode in strip sym; do for m in 1000 10000 30000; do
  if [ $mode = strip ]; then export CLIPPER_HIP_AFFINITY=strip; else unset CLIPPER_HIP_AFFINITY; fi
  timeout 300 python bench.py --m $m --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$mode', d['config']['m'], 'step', d['ms_per_step'], 'aff', d['affinity_ms'], 'aff_kernel', d['affinity_kernel_ms'], 'solve', d['solve_ms'], 'score', d['solution']['score'])"
done; done 2>&1 | tee $OUT/ab.log
