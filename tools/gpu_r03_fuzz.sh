#!/bin/bash
# extended run of the row-view / DSD fuzz (tests/test_gpu_fuzz_views.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
FUZZ_VIEWS_CASES=${1:-60} FUZZ_DSD_CASES=${2:-40} FUZZ_VIEWS_SEED=${3:-20260927} timeout 400 python -m pytest tests/test_gpu_fuzz_views.py -x -q > gpurun_out/r03j_fuzz_tests.txt 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r03j_fuzz_tests.txt | cut -c1-400
tail -2 gpurun_out/fuzz_views.log | cut -c1-300; tail -1 gpurun_out/fuzz_dsd.log
