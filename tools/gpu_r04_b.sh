#!/bin/bash
# round 4, session B: where the resident solver on a view spends its time
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04b
export CLIPPER_HIP_STAMPS=1
for w in auto 48 96 140 200; do
  if [ "$w" = auto ]; then unset CLIPPER_HIP_VIEW_RESIDENT_WGS; else export CLIPPER_HIP_VIEW_RESIDENT_WGS=$w; fi
  timeout 120 python tools/rvr_timeline.py >> gpurun_out/r04b/rvr_timeline.txt 2>&1
done
unset CLIPPER_HIP_VIEW_RESIDENT_WGS
CLIPPER_HIP_VIEW_RESIDENT=0 timeout 120 python tools/rvr_timeline.py >> gpurun_out/r04b/rvr_timeline.txt 2>&1
unset CLIPPER_HIP_STAMPS
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r04b/prof" -o rvr -- python "$GRAFT_REPO_ROOT/bench.py" --steps 12 --warmup 2 --no-cpu-baseline --probe-m 0 > "$GRAFT_REPO_ROOT/gpurun_out/r04b/bench_prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find gpurun_out/r04b/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r04b/kernel_stats.txt 2>&1
find gpurun_out/r04b -name "*.db" -size +8M -delete
cat gpurun_out/r04b/rvr_timeline.txt
head -12 gpurun_out/r04b/kernel_stats.txt
