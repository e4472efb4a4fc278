#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04e
export CLIPPER_HIP_RESIDENT_DEBUG=1
timeout 150 python -m pytest "tests/test_gpu_rv_resident.py::test_matrix_without_points_and_parameter_variants" -x -q -s -o faulthandler_timeout=50 > gpurun_out/r04e/variants.txt 2>&1
echo "variants rc=$?" | tee gpurun_out/r04e/summary.txt
tail -60 gpurun_out/r04e/variants.txt
