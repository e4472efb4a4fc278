#!/bin/bash
out=gpurun_out/r03c; mkdir -p $out
export TMPDIR=/tmp
timeout 200 python bench.py --steps 20 --warmup 3 > $out/bench.log 2>$out/bench.err; echo "bench rc=$?"
timeout 200 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 --profile > $out/rowview_probe.jsonl 2>&1; echo "probe rc=$?"
timeout 100 python -m pytest tests/test_gpu_rowview.py tests/test_gpu_parity.py -q -x > $out/some_tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/some_tests.txt
