#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04k
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_csc.py tests/test_gpu_rowview.py tests/test_gpu_rv_resident.py -x -q > gpurun_out/r04k/tests.txt 2>&1
echo "tests rc=$?" | tee gpurun_out/r04k/summary.txt
tail -4 gpurun_out/r04k/tests.txt
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --probe-m 0 > gpurun_out/r04k/bench.log 2>&1
grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*\|"affinity_ms": [0-9.]*' gpurun_out/r04k/bench.log | head -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r04k/prof" -o rvr -- python "$GRAFT_REPO_ROOT/bench.py" --steps 12 --warmup 2 --no-cpu-baseline --probe-m 0 > "$GRAFT_REPO_ROOT/gpurun_out/r04k/bench_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/r04k/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r04k/kernel_stats.txt 2>&1
find gpurun_out/r04k -name "*.db" -delete
head -8 gpurun_out/r04k/kernel_stats.txt
