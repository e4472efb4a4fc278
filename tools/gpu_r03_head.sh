#!/bin/bash
# the suite at the round's HEAD, then the extended fuzz under another seed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 480 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03n_gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed" gpurun_out/r03n_gpu_tests.txt | tail -2; grep -n "^FAILED\|^ERROR" gpurun_out/r03n_gpu_tests.txt | head
bash tools/gpu_r03_fuzz.sh 60 40 777
cp gpurun_out/fuzz_views.log gpurun_out/r03n_fuzz_views_seed777.log; cp gpurun_out/fuzz_dsd.log gpurun_out/r03n_fuzz_dsd_seed778.log
