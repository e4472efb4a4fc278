#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04c
timeout 600 python -m pytest tests/test_gpu_rv_resident.py -x -q > gpurun_out/r04c/rv_resident.txt 2>&1
echo "rv_resident rc=$?" | tee gpurun_out/r04c/summary.txt
export CLIPPER_HIP_STAMPS=1
for w in auto 100 220; do
  if [ "$w" = auto ]; then unset CLIPPER_HIP_VIEW_RESIDENT_WGS; else export CLIPPER_HIP_VIEW_RESIDENT_WGS=$w; fi
  timeout 120 python tools/rvr_timeline.py >> gpurun_out/r04c/rvr_timeline.txt 2>&1
done
unset CLIPPER_HIP_VIEW_RESIDENT_WGS
unset CLIPPER_HIP_STAMPS
cat gpurun_out/r04c/rvr_timeline.txt
tail -15 gpurun_out/r04c/rv_resident.txt
