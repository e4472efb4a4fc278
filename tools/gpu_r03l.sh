#!/bin/bash
out=gpurun_out/r03l; mkdir -p $out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_configs.py -q -s -k "cfg5" > $out/cfg5.txt 2>&1; echo "cfg5 rc=$?"; grep -n "m=300000\|passed\|failed\|Error" $out/cfg5.txt | head
timeout 100 python -m pytest tests/test_gpu_parity.py -q -s -k "any_dimension or duplicate" > $out/edge.txt 2>&1; echo "edge rc=$?"; grep -n "rel dscore\|passed\|failed" $out/edge.txt | head
timeout 500 python -m pytest tests -m gpu -q --durations=10 --deselect tests/test_gpu_configs.py::test_cfg5_m300000_on_one_gpu_sampled_rows_and_solve > $out/gpu_tests.txt 2>&1
echo "gpu tests rc=$?"; grep -n "passed\|failed" $out/gpu_tests.txt | tail -2; grep -n "^FAILED\|^ERROR" $out/gpu_tests.txt | head
