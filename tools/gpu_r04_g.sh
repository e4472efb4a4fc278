#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04g
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r04g/gpu_tests.txt 2>&1
echo "suite rc=$?" | tee gpurun_out/r04g/summary.txt
tail -25 gpurun_out/r04g/gpu_tests.txt
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r04g/bench.log 2> gpurun_out/r04g/bench.err
grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r04g/bench.log | head -6
