#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04d
timeout 420 python -m pytest tests/test_gpu_rv_resident.py -x -q --timeout 120 > gpurun_out/r04d/rv_resident.txt 2>&1
echo "rv_resident rc=$?" | tee gpurun_out/r04d/summary.txt
export CLIPPER_HIP_STAMPS=1
timeout 120 python tools/rvr_timeline.py >> gpurun_out/r04d/rvr_timeline.txt 2>&1
unset CLIPPER_HIP_STAMPS
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --probe-m 0 > gpurun_out/r04d/bench.log 2>&1
cat gpurun_out/r04d/rvr_timeline.txt
tail -8 gpurun_out/r04d/rv_resident.txt
grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*' gpurun_out/r04d/bench.log | head -3
