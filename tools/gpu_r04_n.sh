#!/bin/bash
# full GPU suite + smoke + bench at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04n
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r04n/gpu_tests.txt 2>&1
echo "suite rc=$?" | tee gpurun_out/r04n/summary.txt
tail -14 gpurun_out/r04n/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04n/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r04n/summary.txt
timeout 300 python bench.py > gpurun_out/r04n/bench.log 2> gpurun_out/r04n/bench.err
echo "bench rc=$?" | tee -a gpurun_out/r04n/summary.txt
grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*\|"ms_per_step": [0-9.]*\|"ms_per_step_host_buffers": [0-9.]*' gpurun_out/r04n/bench.log | head -8
