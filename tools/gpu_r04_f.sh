#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04f
timeout 300 python -m pytest tests/test_gpu_rv_resident.py -x -q > gpurun_out/r04f/rv_resident.txt 2>&1
echo "rv_resident rc=$?" | tee gpurun_out/r04f/summary.txt
tail -4 gpurun_out/r04f/rv_resident.txt
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > gpurun_out/r04f/bench_ht.log 2> gpurun_out/r04f/host_timing.txt
tail -24 gpurun_out/r04f/host_timing.txt
