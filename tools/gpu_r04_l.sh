#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_csc.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r04l/tests.txt 2>&1
echo "tests rc=$?" | tee gpurun_out/r04l/summary.txt
tail -6 gpurun_out/r04l/tests.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --probe-m 0 --storage csc64 > gpurun_out/r04l/bench_csc64.log 2>&1
grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*\|"affinity_ms": [0-9.]*\|"affinity_kernel_ms": [0-9.]*' gpurun_out/r04l/bench_csc64.log | head -4
timeout 400 python tools/run_configs.py --storage csc64 --configs 10k,100k,300k --reps 2 --no-cpu > gpurun_out/r04l/configs_csc64.jsonl 2>&1; cut -c1-400 gpurun_out/r04l/configs_csc64.jsonl
