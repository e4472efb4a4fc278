#!/bin/bash
tag=${1:-r03c}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q --durations=40 --timeout=40 -p no:cacheprovider \
  --deselect "tests/test_gpu_configs.py::test_north_star_m100000_against_the_golden_oracle_answer" \
  --deselect "tests/test_gpu_configs.py::test_sweep_m30000_against_the_oracle" \
  > $out/gpu_tests.txt 2>&1
echo "gpu tests rc=$?" | tee -a $out/summary.txt
grep -n "passed\|failed" $out/gpu_tests.txt | tail -3
grep -n "^FAILED\|^ERROR" $out/gpu_tests.txt | head -40
grep -n "slowest" -A 42 $out/gpu_tests.txt | head -60
