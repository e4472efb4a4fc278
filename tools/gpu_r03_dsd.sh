#!/bin/bash
# Round 3, third GPU session: the suite at HEAD, then the exact DSD rounding timed at the sweep's sizes
# (tools/dsd_timing.py; CLIPPER_HIP_HOST_TIMING prints gather / host / flows per call).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 480 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_gpu_tests.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r03c_gpu_tests.txt
tail -3 gpurun_out/r03c_gpu_tests.txt
CLIPPER_HIP_HOST_TIMING=1 timeout 300 python tools/dsd_timing.py --sizes 10000,30000,100000 > gpurun_out/r03c_dsd_timing.jsonl 2> gpurun_out/r03c_dsd_timing.err
grep "^\[dsd\]" gpurun_out/r03c_dsd_timing.err > gpurun_out/r03c_dsd_host.txt
cat gpurun_out/r03c_dsd_timing.jsonl; cat gpurun_out/r03c_dsd_host.txt
CLIPPER_HIP_HOST_TIMING=1 timeout 200 python tools/dsd_timing.py --sizes 300000 > gpurun_out/r03c_dsd_timing_300k.jsonl 2> gpurun_out/r03c_dsd_timing_300k.err
grep "^\[dsd\]" gpurun_out/r03c_dsd_timing_300k.err; cat gpurun_out/r03c_dsd_timing_300k.jsonl
