#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --durations=8 > $out/gpu_tests.txt 2>&1
echo "gpu tests rc=$?"; grep -n "passed\|failed" $out/gpu_tests.txt | tail -2; grep -n "^FAILED\|^ERROR" $out/gpu_tests.txt | head
timeout 200 python bench.py --steps 20 --warmup 3 > $out/bench.log 2>$out/bench.err; echo "bench rc=$?"; tail -1 $out/bench.log | cut -c1-3000; tail -3 $out/bench.err
timeout 900 bash tools/gpu_prof_r03.sh r03j_prof 398c3fb
