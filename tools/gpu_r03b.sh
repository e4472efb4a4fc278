#!/bin/bash
tag=${1:-r03b}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $out/gpu_tests.txt 2>&1
echo "gpu tests rc=$?" | tee -a $out/summary.txt
grep -n "passed\|failed" $out/gpu_tests.txt | tail -3
grep -n "^FAILED\|^ERROR" $out/gpu_tests.txt | head -40
