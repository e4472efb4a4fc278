#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04j
timeout 300 python -m pytest tests/test_gpu_rv_resident.py tests/test_gpu_rowview.py -x -q > gpurun_out/r04j/tests.txt 2>&1
echo "tests rc=$?" | tee gpurun_out/r04j/summary.txt
tail -4 gpurun_out/r04j/tests.txt
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > gpurun_out/r04j/bench_ht.log 2> gpurun_out/r04j/host_timing.txt
tail -12 gpurun_out/r04j/host_timing.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r04j/prof" -o rvr -- python "$GRAFT_REPO_ROOT/bench.py" --steps 12 --warmup 2 --no-cpu-baseline --probe-m 0 > "$GRAFT_REPO_ROOT/gpurun_out/r04j/bench_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/r04j/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r04j/kernel_stats.txt 2>&1
find gpurun_out/r04j -name "*.db" -size +8M -delete
head -16 gpurun_out/r04j/kernel_stats.txt
grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*' gpurun_out/r04j/bench_prof.log | head -3
